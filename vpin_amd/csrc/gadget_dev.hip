// gadget_dev.hip -- vPIN's two gadgets built ON the device (gfx950): R1CS instance, witness, is_sat and the
// dense representation SNARK::encode commits to, without the matrices ever existing on the host.
//
// Replaces, inside the reference's timed span (proof_point_mult.rs:24-101, proof_point_add.rs):
//   vPIN_proof_generation/src/point_mult.rs:61-704      gadget + witness synthesis (2 x 128 inversions per op)
//   vPIN_proof_generation/src/point_addition.rs:67-327
//   Spartan/src/lib.rs:138-244                          Instance::new (padding, column remap)
//   Spartan/src/r1csinstance.rs:240-270                 is_sat
//   Spartan/src/sparse_mlpoly.rs:232-265,368-438        AddrTimestamps::new, multi_sparse_to_dense_rep
// An instance of N operations is N shifted copies of one per-operation template (host/gadget_ops.h, a few
// thousand triplets): every derived structure -- CSR, CSC, the push-order triplets, the read/audit
// timestamps of the memory-checking trace -- is a closed form in (operation j, template entry t), so one
// thread per output element writes it straight into HBM.  Witness synthesis runs one thread per operation
// (the 128 double-and-add steps of an operation are sequential); the two inversions of a step share one
// Fermat exponentiation.  Integer/byte work, HBM-write bound: no LDS tiling, no MFMA.
#include <algorithm>
#include <cstring>
#include <memory>
#include <vector>

#include "gadget_dev.h"
#include "host/gadget_ops.h"

namespace vpin {

constexpr int kGB = 256;

__device__ __forceinline__ uint32_t expand_col(uint32_t tc, size_t j, size_t ov, size_t nv_pad) {
  return (tc & kSpecialBit) ? (uint32_t)(nv_pad + (tc & 1u)) : (uint32_t)(ov * j + tc);
}

// ---- CSR / CSC expansion ---------------------------------------------------------------------------

__global__ __launch_bounds__(kGB) void gd_rowptr_kernel(GadgetTmplDev t, size_t oc, size_t n_ops, size_t nrows_pad,
                                                        uint32_t* __restrict__ rowptr) {
  size_t i = (size_t)blockIdx.x * kGB + threadIdx.x;
  if (i > nrows_pad) return;
  const size_t real = oc * n_ops;
  rowptr[i] = i < real ? (uint32_t)((size_t)t.T * (i / oc) + t.rowptr[i % oc]) : (uint32_t)((size_t)t.T * n_ops);
}

__global__ __launch_bounds__(kGB) void gd_csr_kernel(GadgetTmplDev t, size_t ov, size_t n_ops, size_t nv_pad,
                                                     uint32_t* __restrict__ col, fq* __restrict__ val) {
  size_t k = (size_t)blockIdx.x * kGB + threadIdx.x;
  if (k >= (size_t)t.T * n_ops) return;
  const size_t j = k / t.T, p = k % t.T;
  col[k] = expand_col(t.csr_col[p], j, ov, nv_pad);
  fq_store(val + k, fq_load(t.csr_val + p));
}

__global__ __launch_bounds__(kGB) void gd_colptr_kernel(GadgetTmplDev t, size_t ov, size_t n_ops, size_t nv_pad,
                                                        uint32_t* __restrict__ colptr) {
  size_t cidx = (size_t)blockIdx.x * kGB + threadIdx.x;
  if (cidx > 2 * nv_pad) return;
  const size_t rel_total = (size_t)t.Trel * n_ops, real = ov * n_ops;
  size_t v;
  if (cidx < real) v = (size_t)t.Trel * (cidx / ov) + t.colptr[cidx % ov];
  else if (cidx <= nv_pad) v = rel_total;
  else if (cidx == nv_pad + 1) v = rel_total + (size_t)t.S[0] * n_ops;
  else v = rel_total + ((size_t)t.S[0] + t.S[1]) * n_ops;
  colptr[cidx] = (uint32_t)v;
}

__global__ __launch_bounds__(kGB) void gd_csc_kernel(GadgetTmplDev t, size_t oc, size_t n_ops, uint32_t* __restrict__ row,
                                                     fq* __restrict__ val) {
  size_t k = (size_t)blockIdx.x * kGB + threadIdx.x;
  const size_t rel_total = (size_t)t.Trel * n_ops, s0 = (size_t)t.S[0] * n_ops, s1 = (size_t)t.S[1] * n_ops;
  if (k >= rel_total + s0 + s1) return;
  if (k < rel_total) {
    const size_t j = k / t.Trel, p = k % t.Trel;
    row[k] = (uint32_t)(oc * j + t.csc_row[p]);
    fq_store(val + k, fq_load(t.csc_val + p));
    return;
  }
  size_t q = k - rel_total;
  const int s = q < s0 ? 0 : 1;
  if (s) q -= s0;
  const size_t j = q / t.S[s], p = q % t.S[s];
  row[k] = (uint32_t)(oc * j + t.spec_row[s][p]);
  fq_store(val + k, fq_load(t.spec_val[s] + p));
}

// push-order triplets (tests; the prover never needs them)
__global__ __launch_bounds__(kGB) void gd_triplets_kernel(GadgetTmplDev t, size_t oc, size_t ov, size_t n_ops, size_t nv_pad,
                                                          uint32_t* __restrict__ row, uint32_t* __restrict__ col,
                                                          fq* __restrict__ val) {
  size_t k = (size_t)blockIdx.x * kGB + threadIdx.x;
  if (k >= (size_t)t.T * n_ops) return;
  const size_t j = k / t.T, p = k % t.T;
  row[k] = (uint32_t)(oc * j + t.row[p]);
  col[k] = expand_col(t.col[p], j, ov, nv_pad);
  fq_store(val + k, fq_load(t.val + p));
}

// ---- SNARK::encode's dense representation -------------------------------------------------------------
// idx layout (spark_dev.h): [row addr A,B,C | row read_ts A,B,C | col addr A,B,C | col read_ts A,B,C] x N, then
// row audit_ts (M), col audit_ts (M).  AddrTimestamps::new (sparse_mlpoly.rs:232-265) walks A, B, C in turn,
// each padded with address 0 up to N entries, with `ts[i] = audit[addr[i]]++`: read_ts = accesses to that
// address so far.  pad_before = padding entries of the earlier matrices (all on address 0); spec_before[s] =
// entries of the earlier matrices in special column s.
struct TracePos {
  size_t pad_before;      // sum over earlier matrices of (N - nnz)
  size_t spec_before[2];  // earlier matrices' entries in special column s (all operations)
  uint32_t row0_upto;     // accesses to row 0 by real entries of matrices 0..m (inclusive)
  uint32_t col0_upto;
};

__global__ __launch_bounds__(kGB) void gd_trace_kernel(GadgetTmplDev t, TracePos tp, int m, size_t oc, size_t ov, size_t n_ops,
                                                       size_t nv_pad, size_t N, uint32_t* __restrict__ idx,
                                                       fq* __restrict__ comb_val) {
  size_t k = (size_t)blockIdx.x * kGB + threadIdx.x;
  if (k >= N) return;
  const size_t nnz = (size_t)t.T * n_ops;
  uint32_t ra, rts, ca, cts;
  if (k < nnz) {
    const size_t j = k / t.T, p = k % t.T;
    const uint32_t ro = t.row[p], tc = t.col[p];
    ra = (uint32_t)(oc * j + ro);
    rts = t.base_row[ro] + t.rank_row[p] + (ra == 0 ? (uint32_t)tp.pad_before : 0u);
    if (tc & kSpecialBit) {
      const int s = tc & 1;
      ca = (uint32_t)(nv_pad + s);
      cts = (uint32_t)(tp.spec_before[s] + (size_t)t.S[s] * j + t.rank_col[p]);
    } else {
      ca = (uint32_t)(ov * j + tc);
      cts = t.base_col[tc] + t.rank_col[p] + (ca == 0 ? (uint32_t)tp.pad_before : 0u);
    }
    fq_store(comb_val + k, fq_load(t.val + p));
  } else {
    ra = ca = 0;
    rts = (uint32_t)(tp.row0_upto + tp.pad_before + (k - nnz));
    cts = (uint32_t)(tp.col0_upto + tp.pad_before + (k - nnz));
    fq_store(comb_val + k, fq_zero());
  }
  idx[(size_t)m * N + k] = ra;
  idx[(size_t)(3 + m) * N + k] = rts;
  idx[(size_t)(6 + m) * N + k] = ca;
  idx[(size_t)(9 + m) * N + k] = cts;
}

__global__ __launch_bounds__(kGB) void gd_audit_kernel(const uint32_t* __restrict__ tot_row, const uint32_t* __restrict__ tot_col,
                                                       size_t oc, size_t ov, size_t n_ops, size_t nv_pad, size_t M, size_t pad_total,
                                                       size_t spec_total0, size_t spec_total1, uint32_t* __restrict__ audit) {
  size_t a = (size_t)blockIdx.x * kGB + threadIdx.x;
  if (a >= M) return;
  uint32_t r = a < oc * n_ops ? tot_row[a % oc] : 0u;
  uint32_t cl = a < ov * n_ops ? tot_col[a % ov] : a == nv_pad ? (uint32_t)spec_total0 : a == nv_pad + 1 ? (uint32_t)spec_total1 : 0u;
  if (a == 0) { r += (uint32_t)pad_total; cl += (uint32_t)pad_total; }
  audit[a] = r;
  audit[M + a] = cl;
}

// ---- witness synthesis ----------------------------------------------------------------------------------

__device__ __noinline__ fq fqm(fq a, fq b) { return fq_mul(a, b); }

// a^(q-2) (zero stays zero, like dalek's Scalar::invert), fixed 4-bit windows
__device__ __noinline__ fq fq_inv(fq a) {
  // q - 2, little-endian 32-bit limbs
  const uint32_t e[8] = {0x5cf5d3ebu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0u, 0u, 0u, 0x10000000u};
  fq tab[16];
  tab[0] = fq_one();
  tab[1] = a;
  for (int i = 2; i < 16; i++) tab[i] = fqm(tab[i - 1], a);
  fq acc = tab[1];  // top nibble of q-2 is 1
  for (int nib = 62; nib >= 0; nib--) {
    acc = fqm(acc, acc); acc = fqm(acc, acc); acc = fqm(acc, acc); acc = fqm(acc, acc);
    const uint32_t d = (e[nib >> 3] >> ((nib & 7) * 4)) & 15u;
    if (d) acc = fqm(acc, tab[d]);
  }
  return acc;
}

// R^2 mod q (ristretto255.rs:309-314): raw integer -> Montgomery form by one product
__device__ __forceinline__ fq fq_r2() {
  fq r;
  r.v[0] = 0x449c0f01u; r.v[1] = 0xa40611e3u; r.v[2] = 0x68859347u; r.v[3] = 0xd00e1ba7u;
  r.v[4] = 0x17f5be65u; r.v[5] = 0xceec73d2u; r.v[6] = 0x7c309a3du; r.v[7] = 0x0399411bu;
  return r;
}

// Scalar::from_bytes_mod_order of 32 little-endian bytes (any 256-bit value), Montgomery form
__device__ __forceinline__ fq fq_from_le32(const uint8_t* __restrict__ b) {
  fq t;
#pragma unroll
  for (int i = 0; i < 8; i++)
    t.v[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
  return fqm(t, fq_r2());
}

// both inverses of a step from one exponentiation (Montgomery's trick on two values); a zero operand
// keeps the reference's 0 -> 0 convention
__device__ __forceinline__ void inv_pair(const fq& x, const fq& y, fq& ix, fq& iy) {
  if (fq_is_zero(x) || fq_is_zero(y)) {
    ix = fq_is_zero(x) ? fq_zero() : fq_inv(x);
    iy = fq_is_zero(y) ? fq_zero() : fq_inv(y);
    return;
  }
  fq t = fq_inv(fqm(x, y));
  ix = fqm(t, y);
  iy = fqm(t, x);
}

// point_mult.rs:414-500 (pa :667-686, pd :688-704) as written: one thread per multiplication, one Fermat
// exponentiation per step.  The exact path for every input; gd_mult_witness_fast_kernel covers the inputs whose
// denominators never vanish and leaves the rest (op_list) to this one.
__global__ __launch_bounds__(64) void gd_mult_witness_kernel(const uint8_t* __restrict__ w16, const uint8_t* __restrict__ px,
                                                             const uint8_t* __restrict__ py, size_t n_ops, fq a_pd,
                                                             fq* __restrict__ vars_para, fq* __restrict__ vars_input,
                                                             fq* __restrict__ vars, const uint32_t* __restrict__ op_list) {
  const size_t tid = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (tid >= n_ops) return;
  const size_t j = op_list ? op_list[tid] : tid;
  constexpr size_t n = vpin_gadgets::kMultBits, ov = vpin_gadgets::kMultVars;
  fq* vi = vars_input + ov * j;
  fq* vv = vars + ov * j;
  const fq one = fq_one(), zero = fq_zero();
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint8_t* b = w16 + 16 * j + 4 * i;
    w[i] = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
  }
  {
    fq wt = zero;
    wt.v[0] = w[0]; wt.v[1] = w[1]; wt.v[2] = w[2]; wt.v[3] = w[3];
    wt = fqm(wt, fq_r2());  // Scalar::from(u128)
    fq_store(vars_para + ov * j + n, wt);
    fq_store(vv + n, wt);
  }
  fq ax = fq_from_le32(px + 32 * j), ay = fq_from_le32(py + 32 * j);
  fq bx = zero, by = zero, bz = one;
#define VPIN_W(k, x) do { fq x_ = (x); fq_store(vi + (k), x_); fq_store(vv + (k), x_); } while (0)
  VPIN_W(n + 1, ax); VPIN_W(2 * n + 2, ay);
  VPIN_W(3 * n + 3, zero); VPIN_W(4 * n + 4, zero); VPIN_W(5 * n + 5, one);
  VPIN_W(10 * n + 8, ax); VPIN_W(10 * n + 9, ay);
  for (size_t i = 0; i < n; i++) {
    fq c, cd;
    inv_pair(fq_sub(bx, ax), fq_dbl(ay), c, cd);
    // pa(bx, by, bz, ax, ay)
    const fq nbz1 = fq_sub(one, bz);
    const fq s1 = fqm(fq_sub(by, ay), c), s2 = fqm(s1, s1);
    const fq t1 = fqm(fq_sub(fq_sub(s2, ax), bx), nbz1), t2 = fqm(ax, bz), cx = fq_add(t1, t2);
    const fq s3 = fqm(s1, fq_sub(ax, cx)), t3 = fqm(fq_sub(s3, ay), nbz1), t4 = fqm(ay, bz), cy = fq_add(t3, t4);
    // pd(ax, ay, a)
    const fq u1 = fqm(ax, ax);
    const fq v1 = fqm(fq_add(fq_add(fq_dbl(u1), u1), a_pd), cd), v2 = fqm(v1, v1);
    const fq dx = fq_sub(v2, fq_dbl(ax)), u2 = fqm(v1, fq_sub(ax, dx)), dy = fq_sub(u2, ay);
    const bool bit = (w[i >> 5] >> (i & 31)) & 1u;
    // products with b / 1-b, b in {0,1}: exact selections
    const fq z1 = bit ? cx : zero, z2 = bit ? zero : bx, z3 = bit ? cy : zero, z4 = bit ? zero : by;
    const fq nbx = fq_add(z1, z2), nby = fq_add(z3, z4), nbz = bit ? zero : bz;
    VPIN_W(i, bit ? one : zero);
    VPIN_W(n + 2 + i, dx); VPIN_W(2 * n + 3 + i, dy);
    VPIN_W(3 * n + 4 + i, nbx); VPIN_W(4 * n + 5 + i, nby); VPIN_W(5 * n + 6 + i, nbz);
    VPIN_W(6 * n + 6 + i, cx); VPIN_W(7 * n + 6 + i, cy); VPIN_W(8 * n + 6 + i, dx); VPIN_W(9 * n + 6 + i, dy);
    VPIN_W(10 * n + 10 + i, c); VPIN_W(11 * n + 10 + i, s1); VPIN_W(12 * n + 10 + i, s2); VPIN_W(13 * n + 10 + i, s3);
    VPIN_W(14 * n + 10 + i, t1); VPIN_W(15 * n + 10 + i, t2); VPIN_W(16 * n + 10 + i, t3); VPIN_W(17 * n + 10 + i, t4);
    VPIN_W(18 * n + 10 + i, cd); VPIN_W(19 * n + 10 + i, u1); VPIN_W(20 * n + 10 + i, v1); VPIN_W(21 * n + 10 + i, v2);
    VPIN_W(22 * n + 10 + i, u2);
    VPIN_W(23 * n + 10 + i, z1); VPIN_W(24 * n + 10 + i, z2); VPIN_W(25 * n + 10 + i, z3); VPIN_W(26 * n + 10 + i, z4);
    ax = dx; ay = dy; bx = nbx; by = nby; bz = nbz;
  }
  VPIN_W(10 * n + 6, bx);
  VPIN_W(10 * n + 7, by);
}


// ---- the same witness with four batched inversions per multiplication instead of 128 ---------------------------
// The doubling chain A_i = 2^i P and the sums C_i = B_i + A_i are first computed in Jacobian coordinates (the
// textbook formulas, which are identities of the affine chord / tangent formulas the gadget encodes: no use of the
// curve equation, so they agree with the reference for any input whose denominators are non-zero), their Z's are
// inverted together (Montgomery's trick, sequential inside the thread), and the two inverse witnesses of every step,
// 1/(bx - ax) and 1/(2 ay), come from one more batch each.  ~8k field multiplications per operation instead of ~42k.
// A vanishing denominator or Z (y = 0 on the doubling chain, B = +-A, x = 0 while B is the point at infinity: inputs
// no honest witness produces) makes the results depend on the reference's inverse-of-zero = 0 convention; such an
// operation is flagged and redone by gd_mult_witness_kernel.  Scratch is slot-major (slot * n_ops + op): coalesced.
struct JacScratch {
  fq* s;
  size_t n_ops, j;
  __device__ __forceinline__ fq ld(size_t slot) const { return fq_load(s + slot * n_ops + j); }
  __device__ __forceinline__ void st(size_t slot, const fq& v) const { fq_store(s + slot * n_ops + j, v); }
};

// v[base .. base+cnt) := their inverses, using prefix slots [pre, pre+cnt); false when one of them is zero
__device__ __noinline__ bool batch_invert(const JacScratch& sc, size_t base, size_t pre, size_t cnt) {
  fq run = sc.ld(base);
  sc.st(pre, run);
  for (size_t i = 1; i < cnt; i++) {
    run = fqm(run, sc.ld(base + i));
    sc.st(pre + i, run);
  }
  if (fq_is_zero(run)) return false;
  fq inv = fq_inv(run);
  for (size_t i = cnt - 1; i >= 1; i--) {
    const fq v = sc.ld(base + i);
    sc.st(base + i, fqm(inv, sc.ld(pre + i - 1)));
    inv = fqm(inv, v);
  }
  sc.st(base, inv);
  return true;
}

constexpr size_t kJacSlots = 4 * (vpin_gadgets::kMultBits + 1) + 3 * vpin_gadgets::kMultBits + 2 * vpin_gadgets::kMultBits;

__global__ __launch_bounds__(64) void gd_mult_witness_fast_kernel(const uint8_t* __restrict__ w16, const uint8_t* __restrict__ px,
                                                                  const uint8_t* __restrict__ py, size_t n_ops, fq a_pd,
                                                                  fq* __restrict__ vars_para, fq* __restrict__ vars_input,
                                                                  fq* __restrict__ vars, fq* __restrict__ scratch,
                                                                  uint32_t* __restrict__ redo_count, uint32_t* __restrict__ redo_list) {
  const size_t j = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (j >= n_ops) return;
  constexpr size_t n = vpin_gadgets::kMultBits, ov = vpin_gadgets::kMultVars;
  // scratch slots: Jacobian A_i (X, Y, Z: n+1 each), prefix products (2n), Jacobian C_i (X, Y, Z: n each), denominators (2n)
  constexpr size_t AX = 0, AY = n + 1, AZ = 2 * (n + 1), PRE = 3 * (n + 1), CX = PRE + 2 * n, CY = CX + n, CZ = CY + n, DEN = CZ + n;
  static_assert(DEN + 2 * n <= kJacSlots + (n + 1), "scratch layout");
  const JacScratch sc{scratch, n_ops, j};
  fq* vi = vars_input + ov * j;
  fq* vv = vars + ov * j;
  const fq one = fq_one(), zero = fq_zero();
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint8_t* b = w16 + 16 * j + 4 * i;
    w[i] = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
  }
  bool ok = true;
  // ---- A: doubling chain in Jacobian coordinates, then affine through one inversion
  {
    fq X = fq_from_le32(px + 32 * j), Y = fq_from_le32(py + 32 * j), Z = one;
    sc.st(AX, X); sc.st(AY, Y); sc.st(AZ, Z);
    for (size_t i = 1; i <= n; i++) {
      const fq XX = fqm(X, X), YY = fqm(Y, Y), YYYY = fqm(YY, YY), ZZ = fqm(Z, Z);
      fq S = fqm(X, YY);
      S = fq_dbl(fq_dbl(S));
      const fq M = fq_add(fq_add(fq_dbl(XX), XX), fqm(a_pd, fqm(ZZ, ZZ)));
      const fq X3 = fq_sub(fqm(M, M), fq_dbl(S));
      const fq Y3 = fq_sub(fqm(M, fq_sub(S, X3)), fq_dbl(fq_dbl(fq_dbl(YYYY))));
      const fq Z3 = fq_dbl(fqm(Y, Z));
      X = X3; Y = Y3; Z = Z3;
      sc.st(AX + i, X); sc.st(AY + i, Y); sc.st(AZ + i, Z);
    }
    ok = batch_invert(sc, AZ, PRE, n + 1);
    if (ok)
      for (size_t i = 0; i <= n; i++) {
        const fq zi = sc.ld(AZ + i), z2 = fqm(zi, zi);
        sc.st(AX + i, fqm(sc.ld(AX + i), z2));
        sc.st(AY + i, fqm(sc.ld(AY + i), fqm(z2, zi)));
      }
  }
  // ---- C: the sums C_i = B_i + A_i in Jacobian coordinates (C_i = A_i while B is the point at infinity)
  if (ok) {
    fq BX = zero, BY = one, BZ = zero;
    bool inf = true;
    for (size_t i = 0; i < n; i++) {
      const fq ax = sc.ld(AX + i), ay = sc.ld(AY + i);
      fq X3, Y3, Z3;
      if (inf) { X3 = ax; Y3 = ay; Z3 = one; }
      else {
        const fq Z1Z1 = fqm(BZ, BZ), U2 = fqm(ax, Z1Z1), S2 = fqm(ay, fqm(BZ, Z1Z1));
        const fq H = fq_sub(U2, BX), R = fq_sub(S2, BY), HH = fqm(H, H), HHH = fqm(H, HH), V = fqm(BX, HH);
        X3 = fq_sub(fq_sub(fqm(R, R), HHH), fq_dbl(V));
        Y3 = fq_sub(fqm(R, fq_sub(V, X3)), fqm(BY, HHH));
        Z3 = fqm(BZ, H);
      }
      sc.st(CX + i, X3); sc.st(CY + i, Y3); sc.st(CZ + i, Z3);
      if ((w[i >> 5] >> (i & 31)) & 1u) { BX = X3; BY = Y3; BZ = Z3; inf = false; }
    }
    ok = batch_invert(sc, CZ, PRE, n);
    if (ok)
      for (size_t i = 0; i < n; i++) {
        const fq zi = sc.ld(CZ + i), z2 = fqm(zi, zi);
        sc.st(CX + i, fqm(sc.ld(CX + i), z2));
        sc.st(CY + i, fqm(sc.ld(CY + i), fqm(z2, zi)));
      }
  }
  // ---- denominators of every step: bx_i - ax_i (slots DEN+i) and 2 ay_i (slots DEN+n+i), one batch
  if (ok) {
    fq bx = zero;
    for (size_t i = 0; i < n; i++) {
      sc.st(DEN + i, fq_sub(bx, sc.ld(AX + i)));
      sc.st(DEN + n + i, fq_dbl(sc.ld(AY + i)));
      if ((w[i >> 5] >> (i & 31)) & 1u) bx = sc.ld(CX + i);
    }
    ok = batch_invert(sc, DEN, PRE, 2 * n);
  }
  if (!ok) {  // an inverse of zero is involved: the step-by-step kernel redoes this operation
    redo_list[atomicAdd(redo_count, 1u)] = (uint32_t)j;
    return;
  }
  // ---- the witness, by the gadget's own formulas (point_mult.rs:414-500) with the inverses at hand
  {
    fq wt = zero;
    wt.v[0] = w[0]; wt.v[1] = w[1]; wt.v[2] = w[2]; wt.v[3] = w[3];
    wt = fqm(wt, fq_r2());  // Scalar::from(u128)
    fq_store(vars_para + ov * j + n, wt);
    fq_store(vv + n, wt);
  }
#define VPIN_W(k, x) do { fq x_ = (x); fq_store(vi + (k), x_); fq_store(vv + (k), x_); } while (0)
  fq ax = sc.ld(AX), ay = sc.ld(AY);
  fq bx = zero, by = zero, bz = one;
  VPIN_W(n + 1, ax); VPIN_W(2 * n + 2, ay);
  VPIN_W(3 * n + 3, zero); VPIN_W(4 * n + 4, zero); VPIN_W(5 * n + 5, one);
  VPIN_W(10 * n + 8, ax); VPIN_W(10 * n + 9, ay);
  for (size_t i = 0; i < n; i++) {
    const fq c = sc.ld(DEN + i), cd = sc.ld(DEN + n + i);
    const fq nbz1 = fq_sub(one, bz);
    const fq s1 = fqm(fq_sub(by, ay), c), s2 = fqm(s1, s1);
    const fq t1 = fqm(fq_sub(fq_sub(s2, ax), bx), nbz1), t2 = fqm(ax, bz), cx = fq_add(t1, t2);
    const fq s3 = fqm(s1, fq_sub(ax, cx)), t3 = fqm(fq_sub(s3, ay), nbz1), t4 = fqm(ay, bz), cy = fq_add(t3, t4);
    const fq u1 = fqm(ax, ax);
    const fq v1 = fqm(fq_add(fq_add(fq_dbl(u1), u1), a_pd), cd), v2 = fqm(v1, v1);
    const fq dx = fq_sub(v2, fq_dbl(ax)), u2 = fqm(v1, fq_sub(ax, dx)), dy = fq_sub(u2, ay);
    const bool bit = (w[i >> 5] >> (i & 31)) & 1u;
    const fq z1 = bit ? cx : zero, z2 = bit ? zero : bx, z3 = bit ? cy : zero, z4 = bit ? zero : by;
    const fq nbx = fq_add(z1, z2), nby = fq_add(z3, z4), nbz = bit ? zero : bz;
    VPIN_W(i, bit ? one : zero);
    VPIN_W(n + 2 + i, dx); VPIN_W(2 * n + 3 + i, dy);
    VPIN_W(3 * n + 4 + i, nbx); VPIN_W(4 * n + 5 + i, nby); VPIN_W(5 * n + 6 + i, nbz);
    VPIN_W(6 * n + 6 + i, cx); VPIN_W(7 * n + 6 + i, cy); VPIN_W(8 * n + 6 + i, dx); VPIN_W(9 * n + 6 + i, dy);
    VPIN_W(10 * n + 10 + i, c); VPIN_W(11 * n + 10 + i, s1); VPIN_W(12 * n + 10 + i, s2); VPIN_W(13 * n + 10 + i, s3);
    VPIN_W(14 * n + 10 + i, t1); VPIN_W(15 * n + 10 + i, t2); VPIN_W(16 * n + 10 + i, t3); VPIN_W(17 * n + 10 + i, t4);
    VPIN_W(18 * n + 10 + i, cd); VPIN_W(19 * n + 10 + i, u1); VPIN_W(20 * n + 10 + i, v1); VPIN_W(21 * n + 10 + i, v2);
    VPIN_W(22 * n + 10 + i, u2);
    VPIN_W(23 * n + 10 + i, z1); VPIN_W(24 * n + 10 + i, z2); VPIN_W(25 * n + 10 + i, z3); VPIN_W(26 * n + 10 + i, z4);
    ax = dx; ay = dy; bx = nbx; by = nby; bz = nbz;
  }
  VPIN_W(10 * n + 6, bx);
  VPIN_W(10 * n + 7, by);
#undef VPIN_W
}
#define VPIN_W(k, x) do { fq x_ = (x); fq_store(vi + (k), x_); fq_store(vv + (k), x_); } while (0)

// point_addition.rs:207-222: one thread per addition
__global__ __launch_bounds__(64) void gd_add_witness_kernel(const uint8_t* __restrict__ px_b, const uint8_t* __restrict__ py_b,
                                                            const uint8_t* __restrict__ rx_b, const uint8_t* __restrict__ ry_b,
                                                            const uint8_t* __restrict__ rz_b, size_t n_ops,
                                                            fq* __restrict__ vars_input, fq* __restrict__ vars) {
  const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= n_ops) return;
  fq* vi = vars_input + vpin_gadgets::kAddVars * i;
  fq* vv = vars + vpin_gadgets::kAddVars * i;
  const fq one = fq_one(), zero = fq_zero();
  const fq px = fq_from_le32(px_b + 32 * i), py = fq_from_le32(py_b + 32 * i);
  const fq rx = fq_from_le32(rx_b + 32 * i), ry = fq_from_le32(ry_b + 32 * i);
  const fq rz = rz_b[i] ? one : zero, nrz = fq_sub(one, rz);
  const fq d = fq_sub(rx, px);
  const fq c = fq_is_zero(d) ? zero : fq_inv(d);
  const fq s1 = fqm(fq_sub(ry, py), c), s2 = fqm(s1, s1);
  const fq t1 = fqm(fq_sub(fq_sub(s2, px), rx), nrz), t2 = fqm(px, rz), x3 = fq_add(t1, t2);
  const fq s3 = fqm(s1, fq_sub(px, x3)), t3 = fqm(fq_sub(s3, py), nrz), t4 = fqm(py, rz), y3 = fq_add(t3, t4);
  VPIN_W(0, c); VPIN_W(1, rx); VPIN_W(2, px); VPIN_W(3, ry); VPIN_W(4, py); VPIN_W(5, rz); VPIN_W(6, s1); VPIN_W(7, s2);
  VPIN_W(8, s3); VPIN_W(9, t1); VPIN_W(10, t2); VPIN_W(11, t3); VPIN_W(12, t4); VPIN_W(13, x3); VPIN_W(14, y3);
#undef VPIN_W
}

// R1CSInstance::is_sat (r1csinstance.rs:240-270): counts rows with Az*Bz != Cz
__global__ __launch_bounds__(kGB) void gd_sat_check_kernel(const fq* __restrict__ Az, const fq* __restrict__ Bz,
                                                           const fq* __restrict__ Cz, size_t n, uint32_t* __restrict__ bad) {
  size_t i = (size_t)blockIdx.x * kGB + threadIdx.x;
  if (i >= n) return;
  const fq a = fq_load(Az + i), b = fq_load(Bz + i), cc = fq_load(Cz + i);
  fq p = (fq_is_zero(a) || fq_is_zero(b)) ? fq_zero() : fq_mul(a, b);
  uint32_t diff = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) diff |= p.v[k] ^ cc.v[k];
  if (diff) atomicAdd(bad, 1u);
}

// ---- host side: template analysis + upload ----------------------------------------------------------------

namespace {

using vpin_host::Fq;

struct TmplSink {
  std::vector<uint32_t> row[3], col[3];
  std::vector<Fq> val[3];
  void put(int m, size_t r, size_t c, const Fq& v) { row[m].push_back((uint32_t)r); col[m].push_back((uint32_t)c); val[m].push_back(v); }
  void A(size_t r, size_t c, const Fq& v) { put(0, r, c, v); }
  void B(size_t r, size_t c, const Fq& v) { put(1, r, c, v); }
  void C(size_t r, size_t c, const Fq& v) { put(2, r, c, v); }
};

static size_t next_pow2(size_t x) { size_t p = 1; while (p < x) p <<= 1; return p; }

// blob builder: every array 16-byte aligned inside one device allocation
struct Blob {
  std::vector<uint8_t> bytes;
  size_t add(const void* p, size_t n) {
    size_t off = (bytes.size() + 15) & ~(size_t)15;
    bytes.resize(off + n);
    if (n) memcpy(bytes.data() + off, p, n);
    return off;
  }
  template <class T>
  size_t vec(const std::vector<T>& v) { return add(v.data(), v.size() * sizeof(T)); }
};

// pooled: a service proving trace after trace recycles the blocks without a round trip through the driver
template <class T>
static int dmalloc(vpin_ctx* c, T** p, size_t n) { return dev_alloc(c, (n ? n : 1) * sizeof(T), (void**)p); }

static unsigned blocks(size_t n) { return (unsigned)((n + kGB - 1) / kGB); }

// Builds the device instance from the per-op template in `ts`.
static int build_instance(vpin_ctx* c, const TmplSink& ts, int kind, size_t n_ops, size_t oc, size_t ov, size_t num_inputs,
                          vpin_dev_instance** out) {
  std::unique_ptr<vpin_dev_instance> g(new (std::nothrow) vpin_dev_instance());
  if (!g) return VPIN_ENOMEM;
  g->kind = kind; g->n_ops = n_ops; g->oc = oc; g->ov = ov;
  const size_t num_cons = oc * n_ops, num_vars = ov * n_ops + 1;
  g->num_cons_unpadded = num_cons;
  g->num_vars_unpadded = num_vars;
  g->num_inputs = num_inputs;
  // Instance::new (lib.rs:146-216)
  const size_t nv_pad = next_pow2(std::max(num_vars, num_inputs + 1));
  const size_t nc_pad = num_cons <= 1 ? 2 : next_pow2(num_cons);
  if (2 * nv_pad >= ((size_t)1 << 32) || nc_pad >= ((size_t)1 << 32)) return VPIN_ESHAPE;

  Blob blob;
  struct Off { size_t row, col, val, rank_row, rank_col, rowptr, csr_col, csr_val, colptr, csc_row, csc_val, spec_row[2], spec_val[2], base_row, base_col; } off[3];
  std::vector<uint32_t> tot_row(oc, 0), tot_col(ov, 0);
  uint32_t T[3], Trel[3], S[3][2];
  for (int m = 0; m < 3; m++) {
    const size_t Tm = ts.row[m].size();
    if (Tm * n_ops >= ((size_t)1 << 32)) return VPIN_ESHAPE;
    T[m] = (uint32_t)Tm;
    std::vector<uint32_t> cnt_row(oc, 0), cnt_col(ov, 0), rank_row(Tm), rank_col(Tm);
    uint32_t sc[2] = {0, 0};
    for (size_t t = 0; t < Tm; t++) {
      const uint32_t r = ts.row[m][t], cl = ts.col[m][t];
      if (r >= oc) return VPIN_ESHAPE;
      rank_row[t] = cnt_row[r]++;
      if (cl & kSpecialBit) {
        const uint32_t s = cl & ~kSpecialBit;
        if (s > 1 || (s == 1 && num_inputs == 0)) return VPIN_ESHAPE;
        rank_col[t] = sc[s]++;
      } else {
        if (cl >= ov) return VPIN_ESHAPE;
        rank_col[t] = cnt_col[cl]++;
      }
    }
    S[m][0] = sc[0]; S[m][1] = sc[1];
    Trel[m] = (uint32_t)(Tm - sc[0] - sc[1]);
    for (size_t i = 0; i < ov; i++)
      if (cnt_col[i] > kLongCol) return VPIN_ESHAPE;  // only the special columns may be long
    // stable row sort / column sort of the template
    std::vector<uint32_t> rowptr(oc + 1, 0), colptr(ov + 1, 0);
    for (size_t i = 0; i < oc; i++) rowptr[i + 1] = rowptr[i] + cnt_row[i];
    for (size_t i = 0; i < ov; i++) colptr[i + 1] = colptr[i] + cnt_col[i];
    std::vector<uint32_t> csr_col(Tm), csc_row(Trel[m]), spec_row[2];
    std::vector<Fq> csr_val(Tm), csc_val(Trel[m]), spec_val[2];
    for (size_t t = 0; t < Tm; t++) {
      const uint32_t r = ts.row[m][t], cl = ts.col[m][t];
      csr_col[rowptr[r] + rank_row[t]] = cl;
      csr_val[rowptr[r] + rank_row[t]] = ts.val[m][t];
      if (cl & kSpecialBit) {
        spec_row[cl & 1].push_back(r);
        spec_val[cl & 1].push_back(ts.val[m][t]);
      } else {
        csc_row[colptr[cl] + rank_col[t]] = r;
        csc_val[colptr[cl] + rank_col[t]] = ts.val[m][t];
      }
    }
    Off& o = off[m];
    o.row = blob.vec(ts.row[m]); o.col = blob.vec(ts.col[m]); o.val = blob.vec(ts.val[m]);
    o.rank_row = blob.vec(rank_row); o.rank_col = blob.vec(rank_col);
    o.rowptr = blob.vec(rowptr); o.csr_col = blob.vec(csr_col); o.csr_val = blob.vec(csr_val);
    o.colptr = blob.vec(colptr); o.csc_row = blob.vec(csc_row); o.csc_val = blob.vec(csc_val);
    for (int s = 0; s < 2; s++) { o.spec_row[s] = blob.vec(spec_row[s]); o.spec_val[s] = blob.vec(spec_val[s]); }
    o.base_row = blob.vec(tot_row);  // accesses by the matrices before m
    o.base_col = blob.vec(tot_col);
    for (size_t i = 0; i < oc; i++) tot_row[i] += cnt_row[i];
    for (size_t i = 0; i < ov; i++) tot_col[i] += cnt_col[i];
    g->nnz[m] = Tm * n_ops;
  }
  const size_t off_tot_row = blob.vec(tot_row), off_tot_col = blob.vec(tot_col);

  (void)hipSetDevice(c->device);
  if (dev_alloc(c, blob.bytes.size(), &g->blob)) return VPIN_ENOMEM;
  auto fail = [&](int rc) { vpin_dev_instance_free(c, g.release()); return rc; };
  if (hipMemcpyAsync(g->blob, blob.bytes.data(), blob.bytes.size(), hipMemcpyHostToDevice, c->stream) != hipSuccess) return fail(VPIN_EHIP);
  const uint8_t* base = (const uint8_t*)g->blob;
  for (int m = 0; m < 3; m++) {
    GadgetTmplDev& t = g->tmpl[m];
    const Off& o = off[m];
    t.T = T[m]; t.Trel = Trel[m]; t.S[0] = S[m][0]; t.S[1] = S[m][1];
    t.row = (const uint32_t*)(base + o.row); t.col = (const uint32_t*)(base + o.col); t.val = (const fq*)(base + o.val);
    t.rank_row = (const uint32_t*)(base + o.rank_row); t.rank_col = (const uint32_t*)(base + o.rank_col);
    t.rowptr = (const uint32_t*)(base + o.rowptr); t.csr_col = (const uint32_t*)(base + o.csr_col); t.csr_val = (const fq*)(base + o.csr_val);
    t.colptr = (const uint32_t*)(base + o.colptr); t.csc_row = (const uint32_t*)(base + o.csc_row); t.csc_val = (const fq*)(base + o.csc_val);
    for (int s = 0; s < 2; s++) { t.spec_row[s] = (const uint32_t*)(base + o.spec_row[s]); t.spec_val[s] = (const fq*)(base + o.spec_val[s]); }
    t.base_row = (const uint32_t*)(base + o.base_row); t.base_col = (const uint32_t*)(base + o.base_col);
  }
  g->tot_row = (const uint32_t*)(base + off_tot_row);
  g->tot_col = (const uint32_t*)(base + off_tot_col);

  // the device R1CS instance (r1cs_dev.h): CSR for the SpMV, CSC for the eval table
  vpin_r1cs_dev* d = new (std::nothrow) vpin_r1cs_dev();
  if (!d) return fail(VPIN_ENOMEM);
  g->r1cs = d;
  d->pooled = true;
  d->owner = c;
  d->num_cons = nc_pad; d->num_vars = nv_pad; d->num_inputs = num_inputs;
  const size_t ncols = 2 * nv_pad;
  for (int m = 0; m < 3; m++) {
    const GadgetTmplDev& t = g->tmpl[m];
    const size_t nnz = g->nnz[m];
    d->nnz[m] = nnz;
    if (dmalloc(c, &d->rowptr[m], nc_pad + 1) || dmalloc(c, &d->csr_col[m], nnz) || dmalloc(c, &d->csr_val[m], nnz) ||
        dmalloc(c, &d->colptr[m], ncols + 1) || dmalloc(c, &d->csc_row[m], nnz) || dmalloc(c, &d->csc_val[m], nnz))
      return fail(VPIN_ENOMEM);
    hipLaunchKernelGGL(gd_rowptr_kernel, dim3(blocks(nc_pad + 1)), dim3(kGB), 0, c->stream, t, oc, n_ops, nc_pad, d->rowptr[m]);
    hipLaunchKernelGGL(gd_colptr_kernel, dim3(blocks(ncols + 1)), dim3(kGB), 0, c->stream, t, ov, n_ops, nv_pad, d->colptr[m]);
    if (nnz) {
      hipLaunchKernelGGL(gd_csr_kernel, dim3(blocks(nnz)), dim3(kGB), 0, c->stream, t, ov, n_ops, nv_pad, d->csr_col[m], d->csr_val[m]);
      hipLaunchKernelGGL(gd_csc_kernel, dim3(blocks(nnz)), dim3(kGB), 0, c->stream, t, oc, n_ops, d->csc_row[m], d->csc_val[m]);
    }
    // long columns: only the special ones can exceed kLongCol entries
    std::vector<uint32_t> longs, lfirst, ck0, ck1;
    size_t first = (size_t)t.Trel * n_ops;
    for (int s = 0; s < 2; s++) {
      const size_t cnt = (size_t)t.S[s] * n_ops;
      if (cnt > kLongCol) {
        longs.push_back((uint32_t)(nv_pad + s));
        lfirst.push_back((uint32_t)ck0.size());
        for (size_t k = first; k < first + cnt; k += kChunk) {
          ck0.push_back((uint32_t)k);
          ck1.push_back((uint32_t)std::min(k + kChunk, first + cnt));
        }
      }
      first += cnt;
    }
    lfirst.push_back((uint32_t)ck0.size());
    d->n_long[m] = longs.size();
    d->n_chunks[m] = ck0.size();
    auto upv = [&](uint32_t** dst, const std::vector<uint32_t>& v) {
      if (dmalloc(c, dst, v.size())) return (int)VPIN_ENOMEM;
      if (!v.empty() && hipMemcpyAsync(*dst, v.data(), v.size() * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) return (int)VPIN_EHIP;
      return (int)VPIN_OK;
    };
    int rc;
    if ((rc = upv(&d->long_cols[m], longs)) || (rc = upv(&d->long_first[m], lfirst)) || (rc = upv(&d->chunk_k0[m], ck0)) ||
        (rc = upv(&d->chunk_k1[m], ck1)))
      return fail(rc);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(VPIN_EHIP);  // host vectors die at scope end
  }
  if (hipGetLastError() != hipSuccess) return fail(VPIN_EHIP);

  // assignment tables, zero-filled (the witness kernels write the live entries)
  int rc;
  if ((rc = vpin_table_alloc(c, nv_pad, &g->vars_para)) || (rc = vpin_table_alloc(c, nv_pad, &g->vars_input)) ||
      (rc = vpin_table_alloc(c, nv_pad, &g->vars)))
    return fail(rc);
  *out = g.release();
  return VPIN_OK;
}

static fq fq_of_host(const Fq& x) {
  fq r;
  memcpy(r.v, &x, 32);
  return r;
}

}  // namespace

int gadget_fill_decomm(vpin_ctx* c, const vpin_dev_instance* g, vpin_spark_decomm* d) {
  if (!c || !g || !d || !d->idx || !d->vals) return VPIN_EINVAL;
  const size_t N = d->N, M = d->M, nv_pad = g->r1cs->num_vars;
  for (int m = 0; m < 3; m++)
    if (g->nnz[m] > N) return VPIN_ESHAPE;
  if (g->oc * g->n_ops > M || 2 * nv_pad > M) return VPIN_ESHAPE;
  (void)hipSetDevice(c->device);
  // tot_row / tot_col / base arrays live on the device; row 0 / column 0 counts are needed here: read them back
  uint32_t base_row0[3], base_col0[3], tot0[2];
  for (int m = 0; m < 3; m++) {
    VPIN_HIP_TRY(hipMemcpyAsync(&base_row0[m], g->tmpl[m].base_row, 4, hipMemcpyDeviceToHost, c->stream));
    VPIN_HIP_TRY(hipMemcpyAsync(&base_col0[m], g->tmpl[m].base_col, 4, hipMemcpyDeviceToHost, c->stream));
  }
  VPIN_HIP_TRY(hipMemcpyAsync(&tot0[0], g->tot_row, 4, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipMemcpyAsync(&tot0[1], g->tot_col, 4, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  size_t pad_before = 0, spec_before[2] = {0, 0};
  for (int m = 0; m < 3; m++) {
    TracePos tp;
    tp.pad_before = pad_before;
    tp.spec_before[0] = spec_before[0]; tp.spec_before[1] = spec_before[1];
    // accesses to address 0 by real entries of matrices 0..m = the base of matrix m+1
    tp.row0_upto = m < 2 ? base_row0[m + 1] : tot0[0];
    tp.col0_upto = m < 2 ? base_col0[m + 1] : tot0[1];
    hipLaunchKernelGGL(gd_trace_kernel, dim3(blocks(N)), dim3(kGB), 0, c->stream, g->tmpl[m], tp, m, g->oc, g->ov, g->n_ops, nv_pad,
                       N, d->idx, d->vals + (size_t)m * N);
    pad_before += N - g->nnz[m];
    spec_before[0] += (size_t)g->tmpl[m].S[0] * g->n_ops;
    spec_before[1] += (size_t)g->tmpl[m].S[1] * g->n_ops;
  }
  hipLaunchKernelGGL(gd_audit_kernel, dim3(blocks(M)), dim3(kGB), 0, c->stream, g->tot_row, g->tot_col, g->oc, g->ov, g->n_ops, nv_pad,
                     M, pad_before, spec_before[0], spec_before[1], d->idx + 12 * N);
  VPIN_HIP_TRY(hipGetLastError());
  return VPIN_OK;
}

}  // namespace vpin

using namespace vpin;

extern "C" {

void vpin_dev_instance_free(vpin_ctx* c, vpin_dev_instance* g) {
  if (!g) return;
  if (c) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
  if (g->r1cs) vpin_r1cs_free(c, g->r1cs);
  vpin_table_free(c, g->vars_para);
  vpin_table_free(c, g->vars_input);
  vpin_table_free(c, g->vars);
  if (g->blob) { if (c) dev_free(c, g->blob); else (void)hipFree(g->blob); }
  delete g;
}

// point_addition.rs:67-327 on the device.  px,py,rx,ry: N x 32 little-endian bytes; rz: N bytes (0/1)
int vpin_gadget_point_add_dev(vpin_ctx* c, const uint8_t* px, const uint8_t* py, const uint8_t* rx, const uint8_t* ry,
                              const uint8_t* rz, size_t N, vpin_dev_instance** out) {
  if (!c || !out || N == 0 || !px || !py || !rx || !ry || !rz) return VPIN_EINVAL;
  const vpin_gadgets::Consts K;
  TmplSink ts;
  vpin_gadgets::emit_add_op(ts, 0, 0, kSpecialBit, K);
  vpin_dev_instance* g = nullptr;
  int rc = build_instance(c, ts, 0, N, vpin_gadgets::kAddCons, vpin_gadgets::kAddVars, 0, &g);
  if (rc) return rc;
  DevBuf in(c);
  if (in.alloc(N * 129)) { vpin_dev_instance_free(c, g); return VPIN_ENOMEM; }
  uint8_t* p = (uint8_t*)in.p;
  const uint8_t* src[4] = {px, py, rx, ry};
  hipError_t e = hipSuccess;
  for (int k = 0; k < 4 && e == hipSuccess; k++) e = hipMemcpyAsync(p + 32 * N * k, src[k], 32 * N, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(p + 128 * N, rz, N, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(gd_add_witness_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, c->stream, p, p + 32 * N, p + 64 * N,
                       p + 96 * N, p + 128 * N, N, g->vars_input->d, g->vars->d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // caller buffers
  if (e != hipSuccess) { set_last_error("vpin_gadget_point_add_dev", e); vpin_dev_instance_free(c, g); return VPIN_EHIP; }
  *out = g;
  return VPIN_OK;
}

// point_mult.rs:61-704 on the device, n = 128.  weights: N x 16 bytes (u128 LE); px,py: N x 32 bytes
int vpin_gadget_point_mult_dev(vpin_ctx* c, const uint8_t* weights_le16, const uint8_t* px, const uint8_t* py, size_t N,
                               vpin_dev_instance** out) {
  if (!c || !out || N == 0 || !weights_le16 || !px || !py) return VPIN_EINVAL;
  const vpin_gadgets::Consts K;
  TmplSink ts;
  vpin_gadgets::emit_mult_op(ts, 0, 0, kSpecialBit, K);
  vpin_dev_instance* g = nullptr;
  int rc = build_instance(c, ts, 1, N, vpin_gadgets::kMultCons, vpin_gadgets::kMultVars, 1, &g);
  if (rc) return rc;
  // the public input a (point_mult.rs:341-342)
  Fq t;
  memcpy(&t, vpin_gadgets::kAPdBytes, 32);
  const Fq a_pd = t * Fq::r2();
  memcpy(g->inputs, &a_pd, 32);
  DevBuf in(c);
  if (in.alloc(N * 80)) { vpin_dev_instance_free(c, g); return VPIN_ENOMEM; }
  uint8_t* p = (uint8_t*)in.p;
  hipError_t e = hipMemcpyAsync(p, weights_le16, 16 * N, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(p + 16 * N, px, 32 * N, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(p + 48 * N, py, 32 * N, hipMemcpyHostToDevice, c->stream);
  // fast kernel first (four batched inversions per operation); operations it flags go through the step-by-step kernel
  DevBuf scr(c), redo(c);
  const bool literal_only = getenv("VPIN_WITNESS_LITERAL") != nullptr;
  const size_t scr_slots = kJacSlots + vpin_gadgets::kMultBits + 1;
  uint32_t n_redo = 0;
  if (e == hipSuccess && !literal_only) {
    if (scr.alloc(scr_slots * N * 32) || redo.alloc((N + 1) * 4)) { vpin_dev_instance_free(c, g); return VPIN_ENOMEM; }
    uint32_t* d_cnt = (uint32_t*)redo.p;
    e = hipMemsetAsync(d_cnt, 0, 4, c->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(gd_mult_witness_fast_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, c->stream, p, p + 16 * N, p + 48 * N,
                         N, fq_of_host(a_pd), g->vars_para->d, g->vars_input->d, g->vars->d, (fq*)scr.p, d_cnt, d_cnt + 1);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&n_redo, d_cnt, 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && n_redo) {
      hipLaunchKernelGGL(gd_mult_witness_kernel, dim3((unsigned)((n_redo + 63) / 64)), dim3(64), 0, c->stream, p, p + 16 * N, p + 48 * N,
                         (size_t)n_redo, fq_of_host(a_pd), g->vars_para->d, g->vars_input->d, g->vars->d, (const uint32_t*)(d_cnt + 1));
      e = hipGetLastError();
    }
  } else if (e == hipSuccess) {
    hipLaunchKernelGGL(gd_mult_witness_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, c->stream, p, p + 16 * N, p + 48 * N, N,
                       fq_of_host(a_pd), g->vars_para->d, g->vars_input->d, g->vars->d, (const uint32_t*)nullptr);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // caller buffers
  if (e != hipSuccess) { set_last_error("vpin_gadget_point_mult_dev", e); vpin_dev_instance_free(c, g); return VPIN_EHIP; }
  *out = g;
  return VPIN_OK;
}

const vpin_r1cs_dev* vpin_dev_instance_r1cs(const vpin_dev_instance* g) { return g ? g->r1cs : nullptr; }
const vpin_table* vpin_dev_instance_vars_para(const vpin_dev_instance* g) { return g ? g->vars_para : nullptr; }
const vpin_table* vpin_dev_instance_vars_input(const vpin_dev_instance* g) { return g ? g->vars_input : nullptr; }
const vpin_table* vpin_dev_instance_vars(const vpin_dev_instance* g) { return g ? g->vars : nullptr; }
const uint8_t* vpin_dev_instance_inputs(const vpin_dev_instance* g) { return (g && g->num_inputs) ? g->inputs : nullptr; }
size_t vpin_dev_instance_num_cons_unpadded(const vpin_dev_instance* g) { return g ? g->num_cons_unpadded : 0; }
size_t vpin_dev_instance_num_vars_unpadded(const vpin_dev_instance* g) { return g ? g->num_vars_unpadded : 0; }
size_t vpin_dev_instance_nnz(const vpin_dev_instance* g, int m) { return (g && m >= 0 && m < 3) ? g->nnz[m] : 0; }

// the push-order triplets of matrix m (what the reference's Vec<(usize, usize, [u8; 32])> holds after
// Instance::new's column remap), materialised for inspection; val as Montgomery limbs
int vpin_dev_instance_triplets(vpin_ctx* c, const vpin_dev_instance* g, int m, uint32_t* row_out, uint32_t* col_out, uint8_t* val_out) {
  if (!c || !g || m < 0 || m > 2 || !row_out || !col_out || !val_out) return VPIN_EINVAL;
  const size_t nnz = g->nnz[m];
  if (!nnz) return VPIN_OK;
  (void)hipSetDevice(c->device);
  DevBuf b(c);
  if (b.alloc(nnz * 40)) return VPIN_ENOMEM;
  fq* v = (fq*)b.p;
  uint32_t* r = (uint32_t*)(v + nnz);
  uint32_t* cl = r + nnz;
  hipLaunchKernelGGL(gd_triplets_kernel, dim3(blocks(nnz)), dim3(kGB), 0, c->stream, g->tmpl[m], g->oc, g->ov, g->n_ops,
                     g->r1cs->num_vars, r, cl, v);
  VPIN_HIP_TRY(hipGetLastError());
  VPIN_HIP_TRY(hipMemcpyAsync(row_out, r, nnz * 4, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipMemcpyAsync(col_out, cl, nnz * 4, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipMemcpyAsync(val_out, v, nnz * 32, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return VPIN_OK;
}

// R1CSInstance::is_sat (r1csinstance.rs:240-270) on the device: 1 = satisfied, 0 = not, < 0 = error
int vpin_dev_instance_is_sat(vpin_ctx* c, const vpin_dev_instance* g) {
  if (!c || !g) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  vpin_table *z = nullptr, *abc[3] = {nullptr, nullptr, nullptr};
  int rc = vpin_r1cs_build_z(c, g->r1cs, g->vars, g->num_inputs ? g->inputs : nullptr, &z);
  if (rc) return rc;
  rc = vpin_r1cs_multiply_vec(c, g->r1cs, z, &abc[0], &abc[1], &abc[2]);
  uint32_t bad = 1;
  if (!rc) {
    DevBuf flag(c);
    if (flag.alloc(16)) rc = VPIN_ENOMEM;
    else {
      hipError_t e = hipMemsetAsync(flag.p, 0, 4, c->stream);
      const size_t n = g->r1cs->num_cons;
      if (e == hipSuccess) {
        hipLaunchKernelGGL(gd_sat_check_kernel, dim3(blocks(n)), dim3(kGB), 0, c->stream, abc[0]->d, abc[1]->d, abc[2]->d, n,
                           (uint32_t*)flag.p);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipMemcpyAsync(&bad, flag.p, 4, hipMemcpyDeviceToHost, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      if (e != hipSuccess) { set_last_error("vpin_dev_instance_is_sat", e); rc = VPIN_EHIP; }
    }
  }
  vpin_table_free(c, z);
  for (auto* t : abc) vpin_table_free(c, t);
  return rc ? rc : (bad == 0 ? 1 : 0);
}

}  // extern "C"
