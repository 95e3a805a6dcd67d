// comm.h -- the exchange layer behind vpin_comm (include/vpin_hip.h): one proof over the GPUs of one node, SPMD.
//
// Every rank runs the same protocol loop on the same inputs; the heavy steps are sharded (commitment rows, product
// circuits, slices) and their SMALL results -- 32-byte compressed rows, three scalars per circuit and round -- are
// all-gathered, after which every rank derives the same transcript challenge by itself.  Those results are already on
// the host when they are exchanged (the transcript lives there), so the latency-critical path is a host all-gather
// through shared memory (ranks = processes of one node, or threads of one process in the rehearsal); device buffers
// (the partial vectors of the evaluation proofs) go through RCCL's ncclAllGather on the context's stream when RCCL is
// enabled, and are staged through the host transport otherwise.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <map>
#include <string>

#include "../../include/vpin_hip.h"

struct vpin_ctx;

namespace vpin {

struct CommSeg;  // the shared segment (POSIX shm or heap)

}  // namespace vpin

struct vpin_comm {
  int rank = 0, world = 1;
  int kind = 0;  // 0 shm, 1 local (threads), 2 callbacks
  vpin::CommSeg* seg = nullptr;
  size_t seg_bytes = 0;
  bool seg_owner = false;      // local transport: the handle that frees the heap segment
  std::atomic<int>* seg_refs = nullptr;  // local transport: handles still alive
  char shm_name[128] = {0};
  uint64_t seq = 0;            // collectives issued so far on this handle
  vpin_allgather_fn cb = nullptr;
  void* cb_user = nullptr;
  bool serialize = false, has_token = false;
  double timeout_s = 120.0;
  // statistics (vpin_comm_stats_read)
  vpin_comm_stats st = {};
  struct TagStat { uint64_t n = 0; double busy_s = 0.0, crit_s = 0.0; };
  std::map<std::string, TagStat> tags;  // the same per call site: the section BEFORE a collective is booked on its tag
  double t_last_exit = 0.0;    // wall clock at the end of the previous collective (or at token acquisition)
  // RCCL (dlopen): device all-gathers on the context's stream
  void* nccl = nullptr;        // ncclComm_t
  vpin_ctx* nccl_ctx = nullptr;
  // pinned staging for device all-gathers without RCCL
  void* h_stage = nullptr;
  size_t h_stage_bytes = 0;
};

namespace vpin {

// host all-gather of `bytes` per rank: recv = world x bytes (recv may alias nothing of send)
int comm_allgather(vpin_comm* cm, const void* send, void* recv, size_t bytes, const char* tag = nullptr);
// device all-gather on c->stream (RCCL when enabled, otherwise D2H + host all-gather + H2D); synchronises only in the staged path
int comm_allgather_dev(vpin_comm* cm, vpin_ctx* c, const void* d_send, void* d_recv, size_t bytes);
// the host all-gather as a proof issues it: in the serialized rehearsal the context's queued GPU work is drained first, so
// that no kernel of this rank runs inside another rank's section
int comm_allgather_ctx(vpin_ctx* c, const void* send, void* recv, size_t bytes, const char* tag = nullptr);
// serialized rehearsal only (a no-op otherwise): an empty collective that closes the section running since the previous
// collective and books it on `tag`, so the critical path can be read per step of the protocol
int comm_mark(vpin_ctx* c, const char* tag);
// marks the group dead (shared-memory and local transports): every peer's next wait returns VPIN_ECOMM
void comm_abort(vpin_comm* cm);
// exit path of a collective entry point: a failure that may be this rank's alone takes the group down with it
inline int comm_leave(vpin_comm* cm, int rc) {
  if (cm && (rc == VPIN_ENOMEM || rc == VPIN_EHIP || rc == VPIN_ECOMM)) comm_abort(cm);
  return rc;
}
// contiguous block of `total` items owned by `rank`: [first, first + count)
inline void comm_block(size_t total, int rank, int world, size_t* first, size_t* count) {
  const size_t per = (total + (size_t)world - 1) / (size_t)world;
  const size_t f = per * (size_t)rank < total ? per * (size_t)rank : total;
  *first = f;
  *count = f + per <= total ? per : total - f;
}
// interleaved rows of a commitment: rank r owns rows r, r + world, ..  (row costs are uneven -- zero padding tails, hot
// columns -- and a contiguous split would leave whole ranks idle)
inline size_t comm_strided_count(size_t total, int rank, int world) {
  return (size_t)rank < total ? (total - (size_t)rank + (size_t)world - 1) / (size_t)world : 0;
}
inline size_t comm_block_max(size_t total, int world) { return (total + (size_t)world - 1) / (size_t)world; }

}  // namespace vpin
