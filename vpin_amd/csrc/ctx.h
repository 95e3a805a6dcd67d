// ctx.h -- internal definitions behind the opaque handles of include/vpin_hip.h
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/vpin_hip.h"
#include "fq_dev.h"

namespace vpin {

void set_last_error(const char* what, hipError_t e);

#define VPIN_HIP_TRY(expr)                         \
  do {                                             \
    hipError_t e_ = (expr);                        \
    if (e_ != hipSuccess) {                        \
      ::vpin::set_last_error(#expr, e_);           \
      return VPIN_EHIP;                            \
    }                                              \
  } while (0)

struct ProfRec {
  int kclass;
  double bytes;
  hipEvent_t start, stop;
};

}  // namespace vpin

struct vpin_table {
  vpin::fq* d = nullptr;  // device pointer
  size_t len = 0;         // live length (halves on bind)
  size_t cap = 0;         // allocated length
  bool owned = true;
};

struct vpin_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int num_cus = 256;
  // scratch for block partials of the round reductions and the final 3 scalars
  vpin::fq* d_partials = nullptr;
  size_t partials_cap = 0;  // in fq elements
  vpin::fq* d_out = nullptr;  // 8 fq
  vpin::fq* h_out = nullptr;  // pinned, 8 fq
  // profiling
  bool prof = false;
  std::vector<vpin::ProfRec> recs;
  std::vector<hipEvent_t> free_events;
  vpin_kstat stats[VPIN_K_COUNT] = {};
  // host-side prover state (generator sets per polynomial size), owned by prover.cpp
  void* prover_cache = nullptr;
  void (*prover_cache_free)(vpin_ctx*) = nullptr;
};

namespace vpin {

// RAII-ish helper used by launchers: brackets a launch with events when profiling is on.
struct ProfScope {
  vpin_ctx* ctx;
  int rec = -1;
  ProfScope(vpin_ctx* c, int kclass, double bytes);
  ~ProfScope();
};

inline bool is_pow2(size_t x) { return x && !(x & (x - 1)); }

}  // namespace vpin
