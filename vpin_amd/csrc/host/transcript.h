// host/transcript.h -- Fiat-Shamir transcript of the prover: Merlin 3.0.0 over STROBE-128 /
// Keccak-f[1600], plus SHAKE256 for generator derivation.  The reference uses the merlin and
// sha3 crates (Spartan/src/transcript.rs:19-43, Spartan/src/random.rs:12-31,
// Spartan/src/commitments.rs:21-25); labels are byte-exact, typos included.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "field.h"

namespace vpin_host {

// One body, two instantiations: the chi step's 25 `~b & c` per round are single ANDN instructions with BMI1 (and the
// rotations RORX with BMI2), which plain x86-64 code generation may not use.  An L5-mult proof absorbs ~95 k scalars and
// points (the `a` vectors of the evaluation proofs, the row commitments), a quarter of a permutation each.
__attribute__((always_inline)) inline void keccak_f1600_body(uint64_t a[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
      0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
      0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
      0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
#define VPIN_ROL(v, n) (((v) << (n)) | ((v) >> (64 - (n))))
  // lanes indexed a[x + 5y]; one round fully unrolled (theta, rho+pi into b, chi, iota): the
  // transcript makes one permutation per challenge, ~2000 of them per polynomial commitment
  uint64_t a00 = a[0], a10 = a[1], a20 = a[2], a30 = a[3], a40 = a[4], a01 = a[5], a11 = a[6], a21 = a[7], a31 = a[8],
           a41 = a[9], a02 = a[10], a12 = a[11], a22 = a[12], a32 = a[13], a42 = a[14], a03 = a[15], a13 = a[16],
           a23 = a[17], a33 = a[18], a43 = a[19], a04 = a[20], a14 = a[21], a24 = a[22], a34 = a[23], a44 = a[24];
  for (int round = 0; round < 24; round++) {
    uint64_t c0 = a00 ^ a01 ^ a02 ^ a03 ^ a04, c1 = a10 ^ a11 ^ a12 ^ a13 ^ a14, c2 = a20 ^ a21 ^ a22 ^ a23 ^ a24,
             c3 = a30 ^ a31 ^ a32 ^ a33 ^ a34, c4 = a40 ^ a41 ^ a42 ^ a43 ^ a44;
    uint64_t d0 = c4 ^ VPIN_ROL(c1, 1), d1 = c0 ^ VPIN_ROL(c2, 1), d2 = c1 ^ VPIN_ROL(c3, 1), d3 = c2 ^ VPIN_ROL(c4, 1),
             d4 = c3 ^ VPIN_ROL(c0, 1);
    a00 ^= d0; a01 ^= d0; a02 ^= d0; a03 ^= d0; a04 ^= d0;
    a10 ^= d1; a11 ^= d1; a12 ^= d1; a13 ^= d1; a14 ^= d1;
    a20 ^= d2; a21 ^= d2; a22 ^= d2; a23 ^= d2; a24 ^= d2;
    a30 ^= d3; a31 ^= d3; a32 ^= d3; a33 ^= d3; a34 ^= d3;
    a40 ^= d4; a41 ^= d4; a42 ^= d4; a43 ^= d4; a44 ^= d4;
    // rho + pi: B[y][2x+3y] = rot(A[x][y], r[x][y])
    uint64_t b00 = a00, b13 = VPIN_ROL(a01, 36), b21 = VPIN_ROL(a02, 3), b34 = VPIN_ROL(a03, 41), b42 = VPIN_ROL(a04, 18);
    uint64_t b02 = VPIN_ROL(a10, 1), b10 = VPIN_ROL(a11, 44), b23 = VPIN_ROL(a12, 10), b31 = VPIN_ROL(a13, 45), b44 = VPIN_ROL(a14, 2);
    uint64_t b04 = VPIN_ROL(a20, 62), b12 = VPIN_ROL(a21, 6), b20 = VPIN_ROL(a22, 43), b33 = VPIN_ROL(a23, 15), b41 = VPIN_ROL(a24, 61);
    uint64_t b01 = VPIN_ROL(a30, 28), b14 = VPIN_ROL(a31, 55), b22 = VPIN_ROL(a32, 25), b30 = VPIN_ROL(a33, 21), b43 = VPIN_ROL(a34, 56);
    uint64_t b03 = VPIN_ROL(a40, 27), b11 = VPIN_ROL(a41, 20), b24 = VPIN_ROL(a42, 39), b32 = VPIN_ROL(a43, 8), b40 = VPIN_ROL(a44, 14);
    // chi (names bXY = B[x][y])
    a00 = b00 ^ (~b10 & b20); a10 = b10 ^ (~b20 & b30); a20 = b20 ^ (~b30 & b40); a30 = b30 ^ (~b40 & b00); a40 = b40 ^ (~b00 & b10);
    a01 = b01 ^ (~b11 & b21); a11 = b11 ^ (~b21 & b31); a21 = b21 ^ (~b31 & b41); a31 = b31 ^ (~b41 & b01); a41 = b41 ^ (~b01 & b11);
    a02 = b02 ^ (~b12 & b22); a12 = b12 ^ (~b22 & b32); a22 = b22 ^ (~b32 & b42); a32 = b32 ^ (~b42 & b02); a42 = b42 ^ (~b02 & b12);
    a03 = b03 ^ (~b13 & b23); a13 = b13 ^ (~b23 & b33); a23 = b23 ^ (~b33 & b43); a33 = b33 ^ (~b43 & b03); a43 = b43 ^ (~b03 & b13);
    a04 = b04 ^ (~b14 & b24); a14 = b14 ^ (~b24 & b34); a24 = b24 ^ (~b34 & b44); a34 = b34 ^ (~b44 & b04); a44 = b44 ^ (~b04 & b14);
    a00 ^= RC[round];
  }
#undef VPIN_ROL
  a[0] = a00; a[1] = a10; a[2] = a20; a[3] = a30; a[4] = a40; a[5] = a01; a[6] = a11; a[7] = a21; a[8] = a31; a[9] = a41;
  a[10] = a02; a[11] = a12; a[12] = a22; a[13] = a32; a[14] = a42; a[15] = a03; a[16] = a13; a[17] = a23; a[18] = a33;
  a[19] = a43; a[20] = a04; a[21] = a14; a[22] = a24; a[23] = a34; a[24] = a44;
}
inline void keccak_f1600_plain(uint64_t a[25]) { keccak_f1600_body(a); }
#if defined(__x86_64__)
__attribute__((target("bmi,bmi2"))) inline void keccak_f1600_bmi(uint64_t a[25]) { keccak_f1600_body(a); }
inline void keccak_f1600(uint64_t a[25]) {
  static const bool bmi = __builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2");
  if (bmi) keccak_f1600_bmi(a);
  else keccak_f1600_plain(a);
}
#else
inline void keccak_f1600(uint64_t a[25]) { keccak_f1600_plain(a); }
#endif

// SHAKE256 extendable output (FIPS 202)
class Shake256 {
  uint64_t st_[25] = {};
  size_t pos_ = 0;
  static constexpr size_t RATE = 136;
  uint8_t* bytes() { return reinterpret_cast<uint8_t*>(st_); }

 public:
  void absorb(const uint8_t* in, size_t n) {
    for (size_t i = 0; i < n; i++) {
      bytes()[pos_++] ^= in[i];
      if (pos_ == RATE) { keccak_f1600(st_); pos_ = 0; }
    }
  }
  void finalize() {
    bytes()[pos_] ^= 0x1f;
    bytes()[RATE - 1] ^= 0x80;
    keccak_f1600(st_);
    pos_ = 0;
  }
  void squeeze(uint8_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
      if (pos_ == RATE) { keccak_f1600(st_); pos_ = 0; }
      out[i] = bytes()[pos_++];
    }
  }
};

// Merlin transcript (STROBE-128, rate 166)
class Transcript {
  alignas(8) uint8_t st_[200];
  uint8_t pos_ = 0, pos_begin_ = 0;
  static constexpr int R = 166;
  enum { F_I = 1, F_A = 2, F_C = 4, F_T = 8, F_M = 16, F_K = 32 };

  void run_f() {
    st_[pos_] ^= pos_begin_;
    st_[pos_ + 1] ^= 0x04;
    st_[R + 1] ^= 0x80;
    keccak_f1600(reinterpret_cast<uint64_t*>(st_));
    pos_ = 0;
    pos_begin_ = 0;
  }
  // st_[pos_ ..] ^= d[0 .. k), k <= R - pos_: eight bytes at a time (the 32-byte scalars and points are most of the volume)
  void xor_in(const uint8_t* d, size_t k) {
    uint8_t* s = st_ + pos_;
    size_t i = 0;
    for (; i + 8 <= k; i += 8) {
      uint64_t a, b;
      memcpy(&a, s + i, 8);
      memcpy(&b, d + i, 8);
      a ^= b;
      memcpy(s + i, &a, 8);
    }
    for (; i < k; i++) s[i] ^= d[i];
    pos_ = (uint8_t)(pos_ + k);
  }
  void absorb(const uint8_t* d, size_t n) {
    while (n) {
      const size_t room = (size_t)(R - pos_), k = n < room ? n : room;
      xor_in(d, k);
      d += k;
      n -= k;
      if (pos_ == R) run_f();
    }
  }
  void squeeze(uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) {
      d[i] = st_[pos_];
      st_[pos_++] = 0;
      if (pos_ == R) run_f();
    }
  }
  void begin_op(uint8_t flags, bool more) {
    if (more) return;
    uint8_t old = pos_begin_;
    pos_begin_ = (uint8_t)(pos_ + 1);
    uint8_t hdr[2] = {old, flags};
    absorb(hdr, 2);
    if ((flags & (F_C | F_K)) && pos_ != 0) run_f();
  }
  void meta_ad(const uint8_t* d, size_t n, bool more) { begin_op(F_M | F_A, more); absorb(d, n); }
  void ad(const uint8_t* d, size_t n, bool more) { begin_op(F_A, more); absorb(d, n); }
  void prf(uint8_t* d, size_t n, bool more) { begin_op(F_I | F_A | F_C, more); squeeze(d, n); }
  static void le32(uint8_t b[4], uint32_t v) { for (int i = 0; i < 4; i++) b[i] = (uint8_t)(v >> (8 * i)); }

 public:
  // Transcript::new(label)
  Transcript(const uint8_t* label, size_t n) {
    memset(st_, 0, sizeof st_);
    const uint8_t hdr[6] = {1, R + 2, 1, 0, 1, 96};
    memcpy(st_, hdr, 6);
    memcpy(st_ + 6, "STROBEv1.0.2", 12);
    keccak_f1600(reinterpret_cast<uint64_t*>(st_));
    meta_ad(reinterpret_cast<const uint8_t*>("Merlin v1.0"), 11, false);
    append_message("dom-sep", label, n);
  }
  explicit Transcript(const char* label) : Transcript(reinterpret_cast<const uint8_t*>(label), strlen(label)) {}

  void append_message(const char* label, const uint8_t* msg, size_t n) {
    uint8_t len[4];
    le32(len, (uint32_t)n);
    meta_ad(reinterpret_cast<const uint8_t*>(label), strlen(label), false);
    meta_ad(len, 4, true);
    ad(msg, n, false);
  }
  void append_message(const char* label, const char* msg) { append_message(label, reinterpret_cast<const uint8_t*>(msg), strlen(msg)); }
  void challenge_bytes(const char* label, uint8_t* out, size_t n) {
    uint8_t len[4];
    le32(len, (uint32_t)n);
    meta_ad(reinterpret_cast<const uint8_t*>(label), strlen(label), false);
    meta_ad(len, 4, true);
    prf(out, n, false);
  }

  // ProofTranscript (Spartan/src/transcript.rs:19-43)
  void append_protocol_name(const char* name) { append_message("protocol-name", name); }
  void append_scalar(const char* label, const Fq& s) { uint8_t b[32]; s.to_bytes(b); append_message(label, b, 32); }
  void append_point(const char* label, const uint8_t c[32]) { append_message(label, c, 32); }
  Fq challenge_scalar(const char* label) { uint8_t b[64]; challenge_bytes(label, b, 64); return Fq::from_bytes_wide(b); }
  std::vector<Fq> challenge_vector(const char* label, size_t n) {
    std::vector<Fq> v(n);
    for (size_t i = 0; i < n; i++) v[i] = challenge_scalar(label);
    return v;
  }
  // AppendToTranscript for [Scalar] (transcript.rs:56-64)
  void append_scalars(const char* label, const Fq* v, size_t n) {
    append_message(label, "begin_append_vector");
    for (size_t i = 0; i < n; i++) append_scalar(label, v[i]);
    append_message(label, "end_append_vector");
  }
  // the same from canonical bytes the caller has already made (n x 32, e.g. converted by a thread team: the conversion out of
  // Montgomery form is a third of the cost of absorbing a long vector)
  void append_scalars_bytes(const char* label, const uint8_t* bytes32, size_t n) {
    append_message(label, "begin_append_vector");
    for (size_t i = 0; i < n; i++) append_message(label, bytes32 + 32 * i, 32);
    append_message(label, "end_append_vector");
  }
};

// RandomTape (Spartan/src/random.rs:12-31) with the OsRng draw made an explicit input: seed64
// stands for the 8 x next_u64 that Scalar::random consumes (ristretto255.rs:381-387).
inline Transcript make_tape(const uint8_t* name, size_t name_len, const uint8_t seed64[64]) {
  Transcript t(name, name_len);
  t.append_scalar("init_randomness", Fq::from_bytes_wide(seed64));
  return t;
}

}  // namespace vpin_host
