// r1cs_dev.h -- the device-resident R1CS instance behind the opaque vpin_r1cs_dev handle (filled by
// r1cs.hip from host triplets, or by gadget_dev.hip straight from a per-operation template)
#pragma once
#include "ctx.h"

struct vpin_r1cs_dev {
  size_t num_cons = 0, num_vars = 0, num_inputs = 0;
  size_t nnz[3] = {0, 0, 0};
  // CSR: rowptr[num_cons+1], col, val ; CSC: colptr[2*num_vars+1], row, val (val permuted)
  uint32_t *rowptr[3] = {}, *csr_col[3] = {};
  vpin::fq* csr_val[3] = {};
  uint32_t *colptr[3] = {}, *csc_row[3] = {};
  vpin::fq* csc_val[3] = {};
  // long columns (> kLongCol entries) are cut into chunks of kChunk entries; one workgroup per
  // chunk, then one thread per long column adds its chunk partials
  uint32_t* long_cols[3] = {};   // [n_long] column index
  uint32_t* long_first[3] = {};  // [n_long+1] first chunk of each long column
  uint32_t* chunk_k0[3] = {};    // [n_chunks] first entry of the chunk
  uint32_t* chunk_k1[3] = {};    // [n_chunks] end entry
  size_t n_long[3] = {0, 0, 0}, n_chunks[3] = {0, 0, 0};
  // (no mutable scratch here: the instance is immutable and may be proven from several contexts at once)
  bool pooled = false;  // arrays come from the context pool (dev_alloc) instead of hipMalloc
  vpin_ctx* owner = nullptr;  // the context whose pool they came from: they go back THERE whatever context frees the handle
};

namespace vpin {
constexpr uint32_t kLongCol = 256;
constexpr uint32_t kChunk = 2048;  // entries per workgroup for long columns
}  // namespace vpin
