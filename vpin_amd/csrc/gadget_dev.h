// gadget_dev.h -- device-built gadget instances (gadget_dev.hip), shared with spark.cpp's encode
#pragma once
#include "ctx.h"
#include "r1cs_dev.h"
#include "spark_dev.h"

namespace vpin {

// Per-matrix template of ONE gadget operation, resident on the device.  An instance of N operations is
// N shifted copies: entry t of operation j sits at triplet position j*T + t (the reference's push order),
// row = oc*j + row[t], column = ov*j + col[t] for an ordinary column, nv_pad + s for special column s
// (s = 0: the constant-1 column `num_vars`; s = 1: the public input a).
struct GadgetTmplDev {
  uint32_t T = 0, Trel = 0, S[2] = {0, 0};
  // push order
  const uint32_t* row = nullptr;       // [T] row offset
  const uint32_t* col = nullptr;       // [T] column offset, or kSpecialBit | s
  const fq* val = nullptr;             // [T]
  const uint32_t* rank_row = nullptr;  // [T] earlier entries of this matrix in the same row
  const uint32_t* rank_col = nullptr;  // [T] earlier entries of this matrix in the same column (per op; special: within the op)
  // row-sorted (stable) and column-sorted (stable) views
  const uint32_t* rowptr = nullptr;    // [oc+1]
  const uint32_t* csr_col = nullptr;   // [T]
  const fq* csr_val = nullptr;         // [T]
  const uint32_t* colptr = nullptr;    // [ov+1] over ordinary columns
  const uint32_t* csc_row = nullptr;   // [Trel]
  const fq* csc_val = nullptr;         // [Trel]
  const uint32_t* spec_row[2] = {nullptr, nullptr};  // [S[s]]
  const fq* spec_val[2] = {nullptr, nullptr};
  // memory-trace counts: accesses by the matrices before this one, per row / ordinary column offset
  const uint32_t* base_row = nullptr;  // [oc]
  const uint32_t* base_col = nullptr;  // [ov]
};

constexpr uint32_t kSpecialBit = 0x80000000u;

}  // namespace vpin

struct vpin_dev_instance {
  int kind = 0;  // 0 = point addition, 1 = point multiplication
  size_t n_ops = 0, oc = 0, ov = 0;
  size_t num_cons_unpadded = 0, num_vars_unpadded = 0;
  vpin_r1cs_dev* r1cs = nullptr;
  vpin_table *vars_para = nullptr, *vars_input = nullptr, *vars = nullptr;
  size_t num_inputs = 0;
  uint8_t inputs[32] = {};
  vpin::GadgetTmplDev tmpl[3];
  const uint32_t* tot_row = nullptr;  // [oc] accesses per row offset over A, B, C
  const uint32_t* tot_col = nullptr;  // [ov]
  void* blob = nullptr;               // one device allocation behind every template pointer
  size_t nnz[3] = {0, 0, 0};
};

namespace vpin {

// SNARK::encode's dense representation for a device-built instance: fills d->idx (12N + 2M u32) and the
// val slices d->vals (allocated by the caller) in closed form from the template.
int gadget_fill_decomm(vpin_ctx* c, const vpin_dev_instance* g, vpin_spark_decomm* d);

}  // namespace vpin
