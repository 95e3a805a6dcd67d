// gadget_ops.h -- the constraint pattern of ONE operation of vPIN's two gadgets, as a function of
// (first row r0, first variable v0, index nv of the constant-1 column), emitted into a sink.
//   vPIN_proof_generation/src/point_addition.rs:81-205   10 constraints / 15 variables per addition
//   vPIN_proof_generation/src/point_mult.rs:85-322       27n+8 constraints / 27n+10 variables per multiplication
// The host builder (gadgets.cpp) calls it once per operation; the device builder (gadget_dev.hip) calls it
// once with r0 = v0 = 0 and nv = kSpecialCol to get the per-operation template it replicates on the GPU.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "field.h"

namespace vpin_gadgets {

using vpin_host::Fq;

constexpr size_t kMultBits = 128;                      // n (load_data.rs:62)
constexpr size_t kMultCons = 27 * kMultBits + 8;       // 3464
constexpr size_t kMultVars = 27 * kMultBits + 10;      // 3466
constexpr size_t kAddCons = 10, kAddVars = 15;
constexpr uint32_t kSpecialCol = 0x80000000u;          // template marker: column nv (+1: the public input a)

struct Consts {
  Fq one, m1, two, three, m2;
  std::vector<Fq> pow2;
  Consts() : one(Fq::one()), m1(one.neg()), two(one + one), three(two + one), m2(two.neg()), pow2(kMultBits) {
    pow2[0] = one;
    for (size_t i = 1; i < kMultBits; i++) pow2[i] = pow2[i - 1] + pow2[i - 1];
  }
};

// Sink: A(row, col, val), B(row, col, val), C(row, col, val)
template <class Sink>
inline void emit_add_op(Sink& s, size_t r, size_t v, size_t nv, const Consts& k) {
  const Fq &one = k.one, &m1 = k.m1;
  s.A(r + 0, v + 0, one); s.B(r + 0, v + 1, one); s.B(r + 0, v + 2, m1); s.C(r + 0, nv, one);
  s.A(r + 1, v + 3, one); s.A(r + 1, v + 4, m1); s.B(r + 1, v + 0, one); s.C(r + 1, v + 6, one);
  s.A(r + 2, v + 6, one); s.B(r + 2, v + 6, one); s.C(r + 2, v + 7, one);
  s.A(r + 3, v + 7, one); s.A(r + 3, v + 2, m1); s.A(r + 3, v + 1, m1);
  s.B(r + 3, nv, one); s.B(r + 3, v + 5, m1); s.C(r + 3, v + 9, one);
  s.A(r + 4, v + 2, one); s.B(r + 4, v + 5, one); s.C(r + 4, v + 10, one);
  s.A(r + 5, v + 9, one); s.A(r + 5, v + 10, one); s.B(r + 5, nv, one); s.C(r + 5, v + 13, one);
  s.A(r + 6, v + 6, one); s.B(r + 6, v + 2, one); s.B(r + 6, v + 13, m1); s.C(r + 6, v + 8, one);
  s.A(r + 7, v + 8, one); s.A(r + 7, v + 4, m1); s.B(r + 7, nv, one); s.B(r + 7, v + 5, m1); s.C(r + 7, v + 11, one);
  s.A(r + 8, v + 4, one); s.B(r + 8, v + 5, one); s.C(r + 8, v + 12, one);
  s.A(r + 9, v + 11, one); s.A(r + 9, v + 12, one); s.B(r + 9, nv, one); s.C(r + 9, v + 14, one);
}

template <class Sink>
inline void emit_mult_op(Sink& s, size_t r0, size_t v0, size_t nv, const Consts& k) {
  const size_t n = kMultBits, oc = kMultCons;
  const Fq &one = k.one, &m1 = k.m1, &two = k.two, &three = k.three, &m2 = k.m2;
  for (size_t i = 0; i < n; i++) s.A(r0, v0 + i, k.pow2[i]);
  s.B(r0, nv, one); s.C(r0, v0 + n, one);
  for (size_t i = 1; i <= n; i++) { s.A(r0 + i, v0 + i - 1, one); s.B(r0 + i, v0 + i - 1, one); s.C(r0 + i, v0 + i - 1, one); }
  s.A(r0 + n + 1, v0 + n + 1, one); s.A(r0 + n + 1, v0 + 10 * n + 8, m1); s.B(r0 + n + 1, nv, one);
  s.A(r0 + n + 2, v0 + 2 * n + 2, one); s.A(r0 + n + 2, v0 + 10 * n + 9, m1); s.B(r0 + n + 2, nv, one);
  s.A(r0 + n + 3, v0 + 3 * n + 3, one); s.B(r0 + n + 3, nv, one);
  s.A(r0 + n + 4, v0 + 4 * n + 4, one); s.B(r0 + n + 4, nv, one);
  s.A(r0 + n + 5, v0 + 5 * n + 5, one); s.A(r0 + n + 5, nv, m1); s.B(r0 + n + 5, nv, one);
  for (size_t i = 0; i < n; i++) {
    const size_t r = r0 + i * 26 + n, v = v0 + i;
    // PA (point_mult.rs:127-189)
    s.A(r + 6, v + 10 * n + 10, one); s.B(r + 6, v + 3 * n + 3, one); s.B(r + 6, v + n + 1, m1); s.C(r + 6, nv, one);
    s.A(r + 7, v + 4 * n + 4, one); s.A(r + 7, v + 2 * n + 2, m1); s.B(r + 7, v + 10 * n + 10, one); s.C(r + 7, v + 11 * n + 10, one);
    s.A(r + 8, v + 11 * n + 10, one); s.B(r + 8, v + 11 * n + 10, one); s.C(r + 8, v + 12 * n + 10, one);
    s.A(r + 9, v + 12 * n + 10, one); s.A(r + 9, v + n + 1, m1); s.A(r + 9, v + 3 * n + 3, m1);
    s.B(r + 9, nv, one); s.B(r + 9, v + 5 * n + 5, m1); s.C(r + 9, v + 14 * n + 10, one);
    s.A(r + 10, v + n + 1, one); s.B(r + 10, v + 5 * n + 5, one); s.C(r + 10, v + 15 * n + 10, one);
    s.A(r + 11, v + 14 * n + 10, one); s.A(r + 11, v + 15 * n + 10, one); s.B(r + 11, nv, one); s.C(r + 11, v + 6 * n + 6, one);
    s.A(r + 12, v + 11 * n + 10, one); s.B(r + 12, v + n + 1, one); s.B(r + 12, v + 6 * n + 6, m1); s.C(r + 12, v + 13 * n + 10, one);
    s.A(r + 13, v + 13 * n + 10, one); s.A(r + 13, v + 2 * n + 2, m1); s.B(r + 13, nv, one); s.B(r + 13, v + 5 * n + 5, m1); s.C(r + 13, v + 16 * n + 10, one);
    s.A(r + 14, v + 2 * n + 2, one); s.B(r + 14, v + 5 * n + 5, one); s.C(r + 14, v + 17 * n + 10, one);
    s.A(r + 15, v + 16 * n + 10, one); s.A(r + 15, v + 17 * n + 10, one); s.B(r + 15, nv, one); s.C(r + 15, v + 7 * n + 6, one);
    // PD (point_mult.rs:197-241)
    s.A(r + 16, v + 18 * n + 10, one); s.B(r + 16, v + 2 * n + 2, two); s.C(r + 16, nv, one);
    s.A(r + 17, v + n + 1, one); s.B(r + 17, v + n + 1, one); s.C(r + 17, v + 19 * n + 10, one);
    s.A(r + 18, v + 19 * n + 10, three); s.A(r + 18, nv + 1, one); s.B(r + 18, v + 18 * n + 10, one); s.C(r + 18, v + 20 * n + 10, one);
    s.A(r + 19, v + 20 * n + 10, one); s.B(r + 19, v + 20 * n + 10, one); s.C(r + 19, v + 21 * n + 10, one);
    s.A(r + 20, v + 21 * n + 10, one); s.A(r + 20, v + n + 1, m2); s.B(r + 20, nv, one); s.C(r + 20, v + 8 * n + 6, one);
    s.A(r + 21, v + 20 * n + 10, one); s.B(r + 21, v + n + 1, one); s.B(r + 21, v + 8 * n + 6, m1); s.C(r + 21, v + 22 * n + 10, one);
    s.A(r + 22, v + 22 * n + 10, one); s.A(r + 22, v + 2 * n + 2, m1); s.B(r + 22, nv, one); s.C(r + 22, v + 9 * n + 6, one);
    // bit select (point_mult.rs:247-302)
    s.A(r + 23, v + 6 * n + 6, one); s.B(r + 23, v, one); s.C(r + 23, v + 23 * n + 10, one);
    s.A(r + 24, v + 3 * n + 3, one); s.B(r + 24, nv, one); s.B(r + 24, v, m1); s.C(r + 24, v + 24 * n + 10, one);
    s.A(r + 25, v + 23 * n + 10, one); s.A(r + 25, v + 24 * n + 10, one); s.B(r + 25, nv, one); s.C(r + 25, v + 3 * n + 4, one);
    s.A(r + 26, v + 7 * n + 6, one); s.B(r + 26, v, one); s.C(r + 26, v + 25 * n + 10, one);
    s.A(r + 27, v + 4 * n + 4, one); s.B(r + 27, nv, one); s.B(r + 27, v, m1); s.C(r + 27, v + 26 * n + 10, one);
    s.A(r + 28, v + 25 * n + 10, one); s.A(r + 28, v + 26 * n + 10, one); s.B(r + 28, nv, one); s.C(r + 28, v + 4 * n + 5, one);
    s.A(r + 29, v + 5 * n + 5, one); s.B(r + 29, nv, one); s.B(r + 29, v, m1); s.C(r + 29, v + 5 * n + 6, one);
    s.A(r + 30, v + n + 2, one); s.A(r + 30, v + 8 * n + 6, m1); s.B(r + 30, nv, one);
    s.A(r + 31, v + 2 * n + 3, one); s.A(r + 31, v + 9 * n + 6, m1); s.B(r + 31, nv, one);
  }
  s.A(r0 + oc - 2, v0 + 10 * n + 6, one); s.A(r0 + oc - 2, v0 + 3 * n + 3 + n, m1); s.B(r0 + oc - 2, nv, one);
  s.A(r0 + oc - 1, v0 + 10 * n + 7, one); s.A(r0 + oc - 1, v0 + 4 * n + 4 + n, m1); s.B(r0 + oc - 1, nv, one);
}

// a_pd (point_mult.rs:341-342): the curve coefficient a of E2, little-endian
static const uint8_t kAPdBytes[32] = {157, 27, 50, 101, 63, 42, 38, 142, 68, 159, 245, 15, 16, 47, 75, 58,
                                      203, 87, 15, 3, 219, 183, 77, 94, 64, 118, 147, 233, 124, 16, 184, 7};

}  // namespace vpin_gadgets
