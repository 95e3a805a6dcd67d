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

inline void keccak_f1600(uint64_t a[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
      0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
      0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
      0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  // rho offsets indexed [x + 5y]
  static const int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  auto rotl = [](uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; };
  for (int round = 0; round < 24; round++) {
    uint64_t c[5], d[5], b[25];
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
    for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
    // rho + pi: B[y, 2x+3y] = rot(A[x,y])
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y], RHO[x + 5 * y]);
    for (int y = 0; y < 5; y++)
      for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
    a[0] ^= RC[round];
  }
}

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
  void absorb(const uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) {
      st_[pos_++] ^= d[i];
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
};

// RandomTape (Spartan/src/random.rs:12-31) with the OsRng draw made an explicit input: seed64
// stands for the 8 x next_u64 that Scalar::random consumes (ristretto255.rs:381-387).
inline Transcript make_tape(const uint8_t* name, size_t name_len, const uint8_t seed64[64]) {
  Transcript t(name, name_len);
  t.append_scalar("init_randomness", Fq::from_bytes_wide(seed64));
  return t;
}

}  // namespace vpin_host
