// gadgets.cpp -- vPIN's two R1CS gadgets, witness synthesis and libspartan's instance
// padding, on the host.  C++ counterpart of
//   vPIN_proof_generation/src/point_addition.rs:67-327   (10 constraints / 15 variables per op)
//   vPIN_proof_generation/src/point_mult.rs:61-704       (27n+8 constraints / 27n+10 variables, n = 128)
//   Spartan/src/lib.rs:138-244                           Instance::new (pow-2 padding, column remap)
// plus a synthetic witness generator standing in for the Python inference service
// (src/convolution/Server.py:324-417 writes the same quantities as JSON): points k*G on the
// curve E2 of src/convolution/Client.py:134-143 with k from SplitMix64.
// The 2x128 field inversions per point multiplication are batched with Montgomery's trick
// (the reference inverts one by one, point_mult.rs:671,693).
#include <omp.h>

#include <cstring>
#include <vector>

#include "../../include/vpin_hip.h"
#include "host/field.h"
#include "host/gadget_ops.h"

namespace {

using vpin_host::Fq;

struct Trip {
  std::vector<uint32_t> row, col;
  std::vector<Fq> val;
  void push(size_t r, size_t c, const Fq& v) { row.push_back((uint32_t)r); col.push_back((uint32_t)c); val.push_back(v); }
};

struct TripSink {
  Trip* M;
  void A(size_t r, size_t c, const Fq& v) { M[0].push(r, c, v); }
  void B(size_t r, size_t c, const Fq& v) { M[1].push(r, c, v); }
  void C(size_t r, size_t c, const Fq& v) { M[2].push(r, c, v); }
};

static size_t next_pow2(size_t x) { size_t p = 1; while (p < x) p <<= 1; return p; }

static Fq fq_from_le32(const uint8_t* b) {  // Scalar::from_bytes_mod_order
  Fq t;
  memcpy(t.l, b, 32);
  return t * Fq::r2();
}

// batch inversion (zero stays zero, like dalek's invert on zero)
static void batch_invert(std::vector<Fq>& v) {
  std::vector<Fq> pre(v.size());
  Fq acc = Fq::one();
  for (size_t i = 0; i < v.size(); i++) {
    pre[i] = acc;
    if (!v[i].is_zero()) acc = acc * v[i];
  }
  acc = acc.invert();
  for (size_t i = v.size(); i-- > 0;) {
    if (v[i].is_zero()) continue;
    Fq t = acc * v[i];
    v[i] = acc * pre[i];
    acc = t;
  }
}

}  // namespace

struct vpin_instance {
  size_t num_cons_unpadded = 0, num_vars_unpadded = 0;
  Trip M[3];
  std::vector<Fq> vars_para, vars_input, vars, inputs;
  vpin_r1cs view{};
  void finish(size_t num_cons, size_t num_vars, size_t num_inputs) {
    // Instance::new (lib.rs:146-216): pad to powers of two, remap columns >= num_vars
    num_cons_unpadded = num_cons;
    num_vars_unpadded = num_vars;
    size_t nv_pad = next_pow2(num_vars > num_inputs + 1 ? num_vars : num_inputs + 1);
    size_t nc_pad = (num_cons == 0 || num_cons == 1) ? 2 : next_pow2(num_cons);
    for (int m = 0; m < 3; m++)
      for (auto& c : M[m].col)
        if (c >= num_vars) c += (uint32_t)(nv_pad - num_vars);
    vars_para.resize(nv_pad, Fq::zero());
    vars_input.resize(nv_pad, Fq::zero());
    vars.resize(nv_pad, Fq::zero());
    view.num_cons = nc_pad;
    view.num_vars = nv_pad;
    view.num_inputs = num_inputs;
    for (int m = 0; m < 3; m++) {
      view.nnz[m] = M[m].row.size();
      view.row[m] = M[m].row.data();
      view.col[m] = M[m].col.data();
      view.val[m] = reinterpret_cast<const uint8_t*>(M[m].val.data());
    }
  }
};

extern "C" {

void vpin_instance_free(vpin_instance* g) { delete g; }
const vpin_r1cs* vpin_instance_r1cs(const vpin_instance* g) { return g ? &g->view : nullptr; }
size_t vpin_instance_num_cons_unpadded(const vpin_instance* g) { return g ? g->num_cons_unpadded : 0; }
size_t vpin_instance_num_vars_unpadded(const vpin_instance* g) { return g ? g->num_vars_unpadded : 0; }
const uint8_t* vpin_instance_vars_para(const vpin_instance* g) { return g ? reinterpret_cast<const uint8_t*>(g->vars_para.data()) : nullptr; }
const uint8_t* vpin_instance_vars_input(const vpin_instance* g) { return g ? reinterpret_cast<const uint8_t*>(g->vars_input.data()) : nullptr; }
const uint8_t* vpin_instance_vars(const vpin_instance* g) { return g ? reinterpret_cast<const uint8_t*>(g->vars.data()) : nullptr; }
const uint8_t* vpin_instance_inputs(const vpin_instance* g) { return (g && !g->inputs.empty()) ? reinterpret_cast<const uint8_t*>(g->inputs.data()) : nullptr; }

// R1CSInstance::is_sat (Spartan/src/r1csinstance.rs:240-270): 1 = satisfied
int vpin_instance_is_sat(const vpin_instance* g) {
  if (!g) return VPIN_EINVAL;
  const size_t nv = g->view.num_vars, nc = g->view.num_cons;
  std::vector<Fq> z(2 * nv, Fq::zero());
  memcpy(z.data(), g->vars.data(), nv * 32);
  z[nv] = Fq::one();
  for (size_t i = 0; i < g->inputs.size(); i++) z[nv + 1 + i] = g->inputs[i];
  std::vector<Fq> abc[3];
  for (int m = 0; m < 3; m++) {
    abc[m].assign(nc, Fq::zero());
    for (size_t k = 0; k < g->M[m].row.size(); k++) abc[m][g->M[m].row[k]] = abc[m][g->M[m].row[k]] + g->M[m].val[k] * z[g->M[m].col[k]];
  }
  for (size_t i = 0; i < nc; i++)
    if (!(abc[0][i] * abc[1][i] == abc[2][i])) return 0;
  return 1;
}

// Shape of the instance a gadget call will build, from the number of operations alone: padded num_cons / num_vars
// (Instance::new pads to powers of two) and the non-zero entries of A, B, C -- what sizes the generator sets.  The per-operation
// counts come from emitting one operation, not from a table of constants.
int vpin_gadget_shape(int is_mult, size_t n_ops, size_t* num_cons, size_t* num_vars, size_t nnz[3]) {
  if (!num_cons || !num_vars || !nnz) return VPIN_EINVAL;
  const size_t n = 128, oc = is_mult ? 27 * n + 8 : 10, ov = is_mult ? n + 10 + n * 26 : 15;
  const size_t cons = oc * n_ops, vars = ov * n_ops + 1;
  Trip M[3];
  TripSink sink{M};
  const vpin_gadgets::Consts K;
  if (is_mult) vpin_gadgets::emit_mult_op(sink, 0, 0, ov + 1, K);
  else vpin_gadgets::emit_add_op(sink, 0, 0, ov + 1, K);
  for (int m = 0; m < 3; m++) nnz[m] = M[m].row.size() * n_ops;
  *num_vars = next_pow2(vars > 1 ? vars : 1);
  *num_cons = (cons == 0 || cons == 1) ? 2 : next_pow2(cons);
  return VPIN_OK;
}

// point_addition.rs:67-327.  px,py,rx,ry: N x 32 little-endian bytes; rz: N bytes (0/1)
int vpin_gadget_point_add(const uint8_t* px_b, const uint8_t* py_b, const uint8_t* rx_b, const uint8_t* ry_b,
                          const uint8_t* rz_b, size_t N, vpin_instance** out) {
  if (!out || (N && (!px_b || !py_b || !rx_b || !ry_b || !rz_b))) return VPIN_EINVAL;
  vpin_instance* g = new (std::nothrow) vpin_instance();
  if (!g) return VPIN_ENOMEM;
  const size_t num_cons = 10 * N, num_vars = 15 * N + 1, nv = num_vars;
  const vpin_gadgets::Consts K;
  const Fq one = K.one;
  TripSink sink{g->M};
  for (size_t i = 0; i < N; i++) vpin_gadgets::emit_add_op(sink, 10 * i, 15 * i, nv, K);
  g->vars_input.assign(num_vars, Fq::zero());
  std::vector<Fq> cinv(N);
  std::vector<Fq> px(N), py(N), rx(N), ry(N), rz(N);
  for (size_t i = 0; i < N; i++) {
    px[i] = fq_from_le32(px_b + 32 * i); py[i] = fq_from_le32(py_b + 32 * i);
    rx[i] = fq_from_le32(rx_b + 32 * i); ry[i] = fq_from_le32(ry_b + 32 * i);
    rz[i] = rz_b[i] ? one : Fq::zero();
    cinv[i] = rx[i] - px[i];
  }
  batch_invert(cinv);
  for (size_t i = 0; i < N; i++) {
    Fq c = cinv[i], s1 = (ry[i] - py[i]) * c, s2 = s1 * s1;
    Fq t1 = (s2 - px[i] - rx[i]) * (one - rz[i]), t2 = px[i] * rz[i], x3 = t1 + t2;
    Fq s3 = s1 * (px[i] - x3), t3 = (s3 - py[i]) * (one - rz[i]), t4 = py[i] * rz[i], y3 = t3 + t4;
    Fq* w = g->vars_input.data() + 15 * i;
    w[0] = c; w[1] = rx[i]; w[2] = px[i]; w[3] = ry[i]; w[4] = py[i]; w[5] = rz[i]; w[6] = s1; w[7] = s2; w[8] = s3;
    w[9] = t1; w[10] = t2; w[11] = t3; w[12] = t4; w[13] = x3; w[14] = y3;
  }
  g->vars_para.assign(num_vars, Fq::zero());  // no model parameters in the add gadget (point_addition.rs:223-224)
  g->vars = g->vars_input;
  g->finish(num_cons, num_vars, 0);
  *out = g;
  return VPIN_OK;
}

// point_mult.rs:61-704 with n = 128 (load_data.rs:62).  weights: N x 16 bytes (u128 LE);
// px,py: N x 32 bytes.
int vpin_gadget_point_mult(const uint8_t* weights_le16, const uint8_t* px_b, const uint8_t* py_b, size_t N,
                           vpin_instance** out) {
  if (!out || (N && (!weights_le16 || !px_b || !py_b))) return VPIN_EINVAL;
  vpin_instance* g = new (std::nothrow) vpin_instance();
  if (!g) return VPIN_ENOMEM;
  const size_t n = 128, oc = 27 * n + 8, ov = n + 10 + n * 26;
  const size_t num_cons = oc * N, num_vars = ov * N + 1, nv = num_vars;
  const vpin_gadgets::Consts K;
  const Fq one = K.one, zero = Fq::zero(), two = K.two, three = K.three;
  const std::vector<Fq>& pow2 = K.pow2;
  for (int m = 0; m < 3; m++) {
    size_t per = m == 0 ? 5260 : m == 1 ? 4488 : 3201;
    g->M[m].row.reserve(per * N); g->M[m].col.reserve(per * N); g->M[m].val.reserve(per * N);
  }
  TripSink sink{g->M};
  for (size_t j = 0; j < N; j++) vpin_gadgets::emit_mult_op(sink, oc * j, ov * j, nv, K);

  const Fq a_pd = fq_from_le32(vpin_gadgets::kAPdBytes);

  // witness synthesis (point_mult.rs:414-500, pa :667-686, pd :688-704).  The doubling chain
  // A_i does not depend on the bits, so all N*n doublings are done first with one batched
  // inversion per step; the addition chain B_i then runs per op with its own batch per step.
  g->vars_para.assign(num_vars, zero);
  g->vars_input.assign(num_vars, zero);
  std::vector<Fq> ax(N), ay(N), bx(N, zero), by(N, zero), bz(N, one), inv_pa(N), inv_pd(N);
  std::vector<unsigned __int128> wts(N);
  for (size_t j = 0; j < N; j++) {
    unsigned __int128 w = 0;
    memcpy(&w, weights_le16 + 16 * j, 16);
    wts[j] = w;
    ax[j] = fq_from_le32(px_b + 32 * j);
    ay[j] = fq_from_le32(py_b + 32 * j);
    Fq* vi = g->vars_input.data() + ov * j;
    Fq wlo = Fq::from_u64((uint64_t)w), whi = Fq::from_u64((uint64_t)(w >> 64));
    g->vars_para[ov * j + n] = wlo + whi * pow2[64];  // Scalar::from(u128)
    vi[n + 1] = ax[j]; vi[2 * n + 2] = ay[j];
    vi[3 * n + 3] = zero; vi[4 * n + 4] = zero; vi[5 * n + 5] = one;
    vi[10 * n + 8] = ax[j]; vi[10 * n + 9] = ay[j];
  }
  for (size_t i = 0; i < n; i++) {
    for (size_t j = 0; j < N; j++) { inv_pa[j] = bx[j] - ax[j]; inv_pd[j] = two * ay[j]; }
    batch_invert(inv_pa);
    batch_invert(inv_pd);
#pragma omp parallel for schedule(static) num_threads(8)
    for (long jj = 0; jj < (long)N; jj++) {
      const size_t j = (size_t)jj;
      Fq* vi = g->vars_input.data() + ov * j;
      // pa(bx, by, bz, ax, ay)
      Fq c = inv_pa[j], s1 = (by[j] - ay[j]) * c, s2 = s1 * s1;
      Fq t1 = (s2 - ax[j] - bx[j]) * (one - bz[j]), t2 = ax[j] * bz[j], cx = t1 + t2;
      Fq s3 = s1 * (ax[j] - cx), t3 = (s3 - ay[j]) * (one - bz[j]), t4 = ay[j] * bz[j], cy = t3 + t4;
      // pd(ax, ay, a)
      Fq cd = inv_pd[j], u1 = ax[j] * ax[j], v1 = (three * u1 + a_pd) * cd, v2 = v1 * v1;
      Fq dx = v2 - two * ax[j], u2 = v1 * (ax[j] - dx), dy = u2 - ay[j];
      const bool bit = (wts[j] >> i) & 1;
      Fq b = bit ? one : zero, nb_ = bit ? zero : one;
      Fq z1 = cx * b, z2 = bx[j] * nb_, nbx = z1 + z2, z3 = cy * b, z4 = by[j] * nb_, nby = z3 + z4, nbz = bz[j] * nb_;
      vi[i] = b;
      vi[n + 2 + i] = dx; vi[2 * n + 3 + i] = dy;
      vi[3 * n + 4 + i] = nbx; vi[4 * n + 5 + i] = nby; vi[5 * n + 6 + i] = nbz;
      vi[6 * n + 6 + i] = cx; vi[7 * n + 6 + i] = cy; vi[8 * n + 6 + i] = dx; vi[9 * n + 6 + i] = dy;
      vi[10 * n + 10 + i] = c; vi[11 * n + 10 + i] = s1; vi[12 * n + 10 + i] = s2; vi[13 * n + 10 + i] = s3;
      vi[14 * n + 10 + i] = t1; vi[15 * n + 10 + i] = t2; vi[16 * n + 10 + i] = t3; vi[17 * n + 10 + i] = t4;
      vi[18 * n + 10 + i] = cd; vi[19 * n + 10 + i] = u1; vi[20 * n + 10 + i] = v1; vi[21 * n + 10 + i] = v2;
      vi[22 * n + 10 + i] = u2;
      vi[23 * n + 10 + i] = z1; vi[24 * n + 10 + i] = z2; vi[25 * n + 10 + i] = z3; vi[26 * n + 10 + i] = z4;
      ax[j] = dx; ay[j] = dy; bx[j] = nbx; by[j] = nby; bz[j] = nbz;
    }
  }
  for (size_t j = 0; j < N; j++) {
    Fq* vi = g->vars_input.data() + ov * j;
    vi[10 * n + 6] = bx[j];
    vi[10 * n + 7] = by[j];
  }
  g->vars.resize(num_vars);
  for (size_t k = 0; k < num_vars; k++) g->vars[k] = g->vars_para[k] + g->vars_input[k];
  g->inputs.assign(1, a_pd);
  g->finish(num_cons, num_vars, 1);
  *out = g;
  return VPIN_OK;
}

// ---- synthetic witness inputs: points k*G on E2 (Jacobian double-and-add over F_q) ---------------

static uint64_t splitmix64(uint64_t& st) {
  st += 0x9E3779B97F4A7C15ULL;
  uint64_t z = st;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

int vpin_synthetic_points(uint64_t seed, size_t count, uint8_t* out_x, uint8_t* out_y) {
  if (!out_x || !out_y) return VPIN_EINVAL;
  // E2 generator and coefficient a (src/convolution/Client.py:138-143), little-endian
  static const uint8_t gx_b[32] = {116,167,235,124,157,206,30,33,147,106,53,151,121,80,64,36,114,103,205,165,60,186,101,245,112,171,179,59,107,253,21,10};
  static const uint8_t gy_b[32] = {150,187,135,98,61,175,138,31,190,253,7,168,130,136,1,89,197,124,10,133,91,144,143,179,127,41,18,102,199,50,131,1};
  static const uint8_t a_b[32] = {157, 27, 50, 101, 63, 42, 38, 142, 68, 159, 245, 15, 16, 47, 75, 58, 203, 87, 15, 3, 219, 183, 77, 94, 64, 118, 147, 233, 124, 16, 184, 7};
  const Fq gx = fq_from_le32(gx_b), gy = fq_from_le32(gy_b), a = fq_from_le32(a_b);
  const Fq one = Fq::one();
  std::vector<uint64_t> ks(count);
  uint64_t st = seed;
  for (size_t i = 0; i < count; i++) { ks[i] = splitmix64(st); if (!ks[i]) ks[i] = 1; }
#pragma omp parallel for schedule(static) num_threads(8)
  for (long ii = 0; ii < (long)count; ii++) {
    // Jacobian (X:Y:Z), x = X/Z^2, y = Y/Z^3
    Fq X = Fq::zero(), Y = one, Z = Fq::zero();
    bool inf = true;
    for (int bit = 63; bit >= 0; bit--) {
      if (!inf) {  // double
        Fq YY = Y * Y, S = (X * YY); S = S + S; S = S + S;
        Fq ZZ = Z * Z, M = X * X; M = M + M + M + a * ZZ * ZZ;
        Fq X3 = M * M - S - S, Y4 = YY * YY, t = Y4 + Y4; t = t + t; t = t + t;
        Fq Y3 = M * (S - X3) - t, Z3 = Y * Z; Z3 = Z3 + Z3;
        X = X3; Y = Y3; Z = Z3;
      }
      if ((ks[ii] >> bit) & 1) {
        if (inf) { X = gx; Y = gy; Z = one; inf = false; }
        else {  // mixed add with affine G (never equal / opposite for 64-bit k on a ~2^252 group)
          Fq ZZ = Z * Z, U2 = gx * ZZ, S2 = gy * ZZ * Z, H = U2 - X, Rr = S2 - Y;
          Fq HH = H * H, HHH = HH * H, V = X * HH;
          Fq X3 = Rr * Rr - HHH - V - V, Y3 = Rr * (V - X3) - Y * HHH, Z3 = Z * H;
          X = X3; Y = Y3; Z = Z3;
        }
      }
    }
    Fq zi = Z.invert(), zi2 = zi * zi;
    (X * zi2).to_bytes(out_x + 32 * ii);
    (Y * zi2 * zi).to_bytes(out_y + 32 * ii);
  }
  return VPIN_OK;
}

}  // extern "C"
