// comm.cpp -- vpin_comm: the exchange layer of one proof over the GPUs of one node (include/vpin_hip.h, comm.h).
//
// Three host transports behind one all-gather (flat: every rank publishes its piece in its own slot of a shared segment
// and reads the others'; pieces are double-buffered by collective parity, so no barrier separates two collectives), a
// compute token for rehearsing N ranks on one GPU, and RCCL (dlopen'ed) for device buffers.  The reference has no
// multi-GPU path; what is exchanged and why is described at the call sites (spark.cpp, prover.cpp).
#include "comm.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <errno.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include <rccl/rccl.h>  // types only: every RCCL entry point is resolved with dlsym

#include "ctx.h"

namespace vpin {

constexpr uint32_t kCommMagic = 0x76504e43u;  // "vPNC"
constexpr uint32_t kNoHolder = 0xffffffffu;

struct alignas(64) CommSlotHdr {
  std::atomic<uint64_t> seq;  // last collective whose piece is complete in data[seq & 1]
  double busy_s[2];           // this rank's section before that collective, by the same parity (a fast rank publishes
                              // collective k + 1 while a slow one still reads k)
  // what this rank believes collective k IS, by the same parity: the whole message's size and the call site.  Every rank
  // compares its peers' values with its own before it copies a byte: ranks that took different protocol branches (a
  // per-process environment knob, an asymmetric early return) fail with VPIN_ECOMM at the first collective they disagree
  // on instead of reading each other's slots at the wrong size.
  uint64_t total_bytes[2];
  uint64_t tag_hash[2];
};

struct alignas(64) CommSeg {
  std::atomic<uint32_t> magic;
  uint32_t world;
  uint64_t slot_bytes;
  std::atomic<uint32_t> attached, abort_flag, token, detached;
  std::atomic<uint32_t> serialize;  // ranks that asked for the token mode
  double created_unix_s;            // when rank 0 initialised the segment (a leftover is older than any attach timeout)
  int32_t creator_pid;              // rank 0's process: a segment whose creator is gone belongs to a job that died
  uint64_t creator_pidns;           // inode of rank 0's /proc/self/ns/pid (0: unknown): a pid means nothing in another namespace
};

// inode of this process's PID namespace (0 when /proc is not there)
static uint64_t pid_namespace_id() {
  struct stat st;
  return stat("/proc/self/ns/pid", &st) == 0 ? (uint64_t)st.st_ino : 0;
}

static inline size_t seg_slot_stride(size_t slot_bytes) { return sizeof(CommSlotHdr) + 2 * ((slot_bytes + 63) & ~(size_t)63); }
static inline size_t seg_size(int world, size_t slot_bytes) { return 4096 + (size_t)world * seg_slot_stride(slot_bytes); }
static inline CommSlotHdr* slot_hdr(CommSeg* s, int r) {
  return reinterpret_cast<CommSlotHdr*>(reinterpret_cast<uint8_t*>(s) + 4096 + (size_t)r * seg_slot_stride(s->slot_bytes));
}
static inline uint8_t* slot_data(CommSeg* s, int r, int parity) {
  return reinterpret_cast<uint8_t*>(slot_hdr(s, r)) + sizeof(CommSlotHdr) + (size_t)parity * ((s->slot_bytes + 63) & ~(size_t)63);
}

static inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static double env_timeout() {
  const char* e = getenv("VPIN_COMM_TIMEOUT_S");
  const double v = e ? atof(e) : 0.0;
  return v > 0.0 ? v : 120.0;
}

// bounded spin: pause for a while, then yield (the CPU tests run more ranks than cores)
struct Spinner {
  double deadline;
  long spins = 0;
  bool lazy;  // serialized rehearsal: the waiters of a token sleep, so that W - 1 spinning threads do not take the
              // host cores the one computing rank's OpenMP teams need
  explicit Spinner(double timeout_s, bool lazy_ = false) : deadline(now_s() + timeout_s), lazy(lazy_) {}
  // false when the deadline has passed
  bool step() {
    spins++;
    if (spins < 2000) { __builtin_ia32_pause(); return true; }
    if ((spins & 63) == 0 && now_s() > deadline) return false;
    if (lazy) usleep(30); else sched_yield();
    return true;
  }
};

// FNV-1a of the call-site tag (0 for an untagged collective)
static uint64_t tag_hash_of(const char* tag) {
  if (!tag) return 0;
  uint64_t h = 0xcbf29ce484222325ull;
  for (const char* p = tag; *p; p++) { h ^= (uint8_t)*p; h *= 0x100000001b3ull; }
  return h ? h : 1;
}

static int comm_disagree(vpin_comm* cm, int peer, uint64_t peer_bytes, uint64_t peer_tag, uint64_t bytes, uint64_t tagh, const char* tag) {
  char msg[256];
  snprintf(msg, sizeof msg, "vpin_comm: rank %d and rank %d disagree on collective %llu: %llu bytes tag %016llx (%s) here, %llu bytes tag %016llx there",
           cm->rank, peer, (unsigned long long)cm->seq, (unsigned long long)bytes, (unsigned long long)tagh, tag ? tag : "-",
           (unsigned long long)peer_bytes, (unsigned long long)peer_tag);
  set_last_error(msg, hipErrorUnknown);
  return VPIN_ECOMM;
}

static void seg_init(CommSeg* s, int world, size_t slot_bytes) {
  s->world = (uint32_t)world;
  s->slot_bytes = slot_bytes;
  s->attached.store(0); s->abort_flag.store(0); s->token.store(kNoHolder); s->detached.store(0); s->serialize.store(0);
  for (int r = 0; r < world; r++) { slot_hdr(s, r)->seq.store(0); slot_hdr(s, r)->busy_s[0] = slot_hdr(s, r)->busy_s[1] = 0.0; }
  s->created_unix_s = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
  s->creator_pid = (int32_t)getpid();
  s->creator_pidns = pid_namespace_id();
  s->magic.store(kCommMagic, std::memory_order_release);
}

static int token_acquire(vpin_comm* cm) {
  if (cm->has_token) return VPIN_OK;
  Spinner sp(cm->timeout_s, true);
  for (;;) {
    uint32_t exp = kNoHolder;
    if (cm->seg->token.compare_exchange_weak(exp, (uint32_t)cm->rank, std::memory_order_acquire)) break;
    if (cm->seg->abort_flag.load(std::memory_order_relaxed)) return VPIN_ECOMM;
    if (!sp.step()) { cm->seg->abort_flag.store(1); return VPIN_ECOMM; }
  }
  cm->has_token = true;
  cm->t_last_exit = now_s();
  return VPIN_OK;
}

static void token_release(vpin_comm* cm) {
  if (!cm->has_token) return;
  cm->has_token = false;
  cm->seg->token.store(kNoHolder, std::memory_order_release);
}

// one piece of at most slot_bytes per rank
static int seg_allgather_piece(vpin_comm* cm, const uint8_t* send, uint8_t* recv, size_t bytes, size_t recv_stride, double busy,
                               double* max_busy, uint64_t total_bytes, uint64_t tagh, const char* tag) {
  CommSeg* s = cm->seg;
  const uint64_t k = ++cm->seq;
  const int parity = (int)(k & 1);
  if (bytes) memcpy(slot_data(s, cm->rank, parity), send, bytes);
  slot_hdr(s, cm->rank)->busy_s[parity] = busy;
  slot_hdr(s, cm->rank)->total_bytes[parity] = total_bytes;
  slot_hdr(s, cm->rank)->tag_hash[parity] = tagh;
  slot_hdr(s, cm->rank)->seq.store(k, std::memory_order_release);
  double mb = busy;
  for (int i = 0; i < cm->world; i++) {
    const int r = (cm->rank + i) % cm->world;  // own piece first, then the neighbours in ring order
    if (r != cm->rank) {
      Spinner sp(cm->timeout_s, cm->serialize);
      while (slot_hdr(s, r)->seq.load(std::memory_order_acquire) < k) {
        if (s->abort_flag.load(std::memory_order_relaxed)) return VPIN_ECOMM;
        if (!sp.step()) { s->abort_flag.store(1); return VPIN_ECOMM; }
      }
      const uint64_t pb = slot_hdr(s, r)->total_bytes[parity], pt = slot_hdr(s, r)->tag_hash[parity];
      if (pb != total_bytes || pt != tagh) {
        s->abort_flag.store(1, std::memory_order_release);
        return comm_disagree(cm, r, pb, pt, total_bytes, tagh, tag);
      }
      const double b = slot_hdr(s, r)->busy_s[parity];
      if (b > mb) mb = b;
    }
    if (bytes) memcpy(recv + (size_t)r * recv_stride, slot_data(s, r, parity), bytes);
  }
  *max_busy = mb;
  return VPIN_OK;
}

int comm_allgather(vpin_comm* cm, const void* send, void* recv, size_t bytes, const char* tag) {
  if (!cm || (bytes && (!send || !recv))) return VPIN_EINVAL;
  const double t_in = now_s();
  const double busy = cm->t_last_exit > 0.0 ? t_in - cm->t_last_exit : 0.0;
  int rc = VPIN_OK;
  double max_busy = busy;
  const uint64_t tagh = tag_hash_of(tag);
  if (cm->world == 1) {
    if (bytes) memcpy(recv, send, bytes);
  } else if (cm->kind == 2) {
    // the caller's fabric: the section time, the size and the call site ride in front of the payload (a fabric that
    // survives ranks sending different sizes hands the mismatch back here; one that does not fails in the callback)
    constexpr size_t kHdr = 24;
    std::vector<uint8_t> sb(kHdr + bytes), rb((size_t)cm->world * (kHdr + bytes));
    const uint64_t b64 = bytes;
    memcpy(sb.data(), &busy, 8); memcpy(sb.data() + 8, &b64, 8); memcpy(sb.data() + 16, &tagh, 8);
    if (bytes) memcpy(sb.data() + kHdr, send, bytes);
    rc = cm->cb(cm->cb_user, sb.data(), rb.data(), kHdr + bytes);
    if (rc) rc = VPIN_ECOMM;
    cm->seq++;
    for (int r = 0; r < cm->world && !rc; r++) {
      const uint8_t* pr = rb.data() + (size_t)r * (kHdr + bytes);
      double b;
      uint64_t pb, pt;
      memcpy(&b, pr, 8); memcpy(&pb, pr + 8, 8); memcpy(&pt, pr + 16, 8);
      if (pb != b64 || pt != tagh) { rc = comm_disagree(cm, r, pb, pt, b64, tagh, tag); break; }
      if (b > max_busy) max_busy = b;
      if (bytes) memcpy((uint8_t*)recv + (size_t)r * bytes, pr + kHdr, bytes);
    }
  } else if (cm->seg->abort_flag.load(std::memory_order_relaxed)) {
    rc = VPIN_ECOMM;  // a rank gave up earlier: the group is dead, whatever the slots still hold
  } else {
    const bool tok = cm->serialize;
    if (tok) token_release(cm);
    const size_t piece = cm->seg->slot_bytes;
    size_t off = 0;
    do {
      const size_t n = bytes - off < piece ? bytes - off : piece;
      double mb = 0.0;
      rc = seg_allgather_piece(cm, (const uint8_t*)send + off, (uint8_t*)recv + off, n, bytes, off == 0 ? busy : 0.0, &mb, bytes, tagh, tag);
      if (rc) break;
      if (mb > max_busy) max_busy = mb;
      off += n;
    } while (off < bytes);
    if (!rc && tok) rc = token_acquire(cm);
  }
  const double t_out = now_s();
  cm->st.collectives++;
  cm->st.bytes += (double)bytes;
  cm->st.wait_s += t_out - t_in;
  cm->st.busy_s += busy;
  cm->st.crit_s += max_busy;
  if (tag) {
    auto& ts = cm->tags[tag];
    ts.n++;
    ts.busy_s += busy;
    ts.crit_s += max_busy;
  }
  cm->t_last_exit = t_out;
  return rc;
}

// A rank that leaves a collective call with a failure of its own (out of memory, a HIP error) tells the group, so that the
// peers return VPIN_ECOMM from their next wait instead of running into the timeout.  The group is dead afterwards.
void comm_abort(vpin_comm* cm) {
  if (cm && cm->seg && cm->world > 1) cm->seg->abort_flag.store(1, std::memory_order_release);
}

int comm_mark(vpin_ctx* c, const char* tag) {
  if (!c || !c->comm || !c->comm->serialize) return VPIN_OK;
  return comm_allgather_ctx(c, nullptr, nullptr, 0, tag);
}

int comm_allgather_ctx(vpin_ctx* c, const void* send, void* recv, size_t bytes, const char* tag) {
  if (!c || !c->comm) return VPIN_EINVAL;
  // (a resident persistent tail kernel is waiting for the challenge this very exchange produces: nothing to drain then)
  if (c->comm->serialize && c->tail_rounds == 0) VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return comm_allgather(c->comm, send, recv, bytes, tag);
}

// ---- RCCL through dlopen ------------------------------------------------------------------------------------------

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.h ? &r : nullptr;
  tried = true;
  // a framework in the same process (torch) may already have its copy loaded: use that one
  for (const char* name : {"librccl.so.1", "librccl.so"}) {
    r.h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
    if (r.h) break;
  }
  if (!r.h)
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.h) break;
    }
  if (!r.h) return nullptr;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
  r.AllGather = (decltype(r.AllGather))dlsym(r.h, "ncclAllGather");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) { r.h = nullptr; return nullptr; }
  return &r;
}

int comm_allgather_dev(vpin_comm* cm, vpin_ctx* c, const void* d_send, void* d_recv, size_t bytes) {
  if (!cm || !c || !d_send || !d_recv || bytes == 0) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  if (cm->nccl && cm->nccl_ctx == c && !cm->serialize) {
    Rccl* r = rccl();
    if (!r) return VPIN_ENODEV;
    ncclResult_t e = r->AllGather(d_send, d_recv, bytes, ncclUint8, (ncclComm_t)cm->nccl, c->stream);
    if (e != ncclSuccess) {
      set_last_error(r->GetErrorString ? r->GetErrorString(e) : "ncclAllGather", hipErrorUnknown);
      return VPIN_ECOMM;
    }
    cm->st.collectives++;
    cm->st.bytes += (double)bytes;
    return VPIN_OK;
  }
  // staged: D2H, host all-gather, H2D (also the path of the serialized rehearsal, whose ranks share one GPU)
  const size_t need = bytes * ((size_t)cm->world + 1);
  if (cm->h_stage_bytes < need) {
    if (cm->h_stage) (void)hipHostFree(cm->h_stage);
    cm->h_stage = nullptr;
    cm->h_stage_bytes = 0;
    VPIN_HIP_TRY(hipHostMalloc(&cm->h_stage, need, hipHostMallocDefault));
    cm->h_stage_bytes = need;
  }
  uint8_t* hs = (uint8_t*)cm->h_stage;
  VPIN_HIP_TRY(hipMemcpyAsync(hs, d_send, bytes, hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  int rc = comm_allgather(cm, hs, hs + bytes, bytes);
  if (rc) return rc;
  VPIN_HIP_TRY(hipMemcpyAsync(d_recv, hs + bytes, bytes * (size_t)cm->world, hipMemcpyHostToDevice, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));  // the staging buffer is reused by the next call
  return VPIN_OK;
}

}  // namespace vpin

using namespace vpin;

extern "C" {

int vpin_comm_create_shm(const char* name, int rank, int world, size_t slot_bytes, vpin_comm** out) {
  if (!name || !out || world < 1 || rank < 0 || rank >= world || name[0] != '/' || strlen(name) >= sizeof(((vpin_comm*)0)->shm_name))
    return VPIN_EINVAL;
  if (slot_bytes == 0) slot_bytes = (size_t)1 << 20;
  const size_t bytes = seg_size(world, slot_bytes);
  const double timeout = env_timeout();
  int fd = -1;
  void* p = MAP_FAILED;
  CommSeg* s = nullptr;
  if (rank == 0) {
    (void)shm_unlink(name);  // a leftover of a job that died with this name
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { if (fd >= 0) close(fd); return VPIN_ECOMM; }
    p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return VPIN_ECOMM;
    s = static_cast<CommSeg*>(p);
    seg_init(s, world, slot_bytes);
  } else {
    // The name may still point at the segment of a job that died with it (rank 0 replaces it, but this rank can get there
    // first): such a leftover passes every check below -- size, magic, world.  So after the segment looks initialised the
    // name is resolved AGAIN, before this rank counts itself in: rank 0 keeps the name until everybody is attached, so at
    // that moment the name must still lead to the very object mapped here.  If it does not (replaced, or gone), this was
    // a leftover: unmap and start over.  (Callers should still make the name unique per job; bench.py does.)
    Spinner sp(timeout);
    for (;;) {
      fd = shm_open(name, O_RDWR, 0600);
      struct stat st_mine;
      if (fd >= 0 && !(fstat(fd, &st_mine) == 0 && (size_t)st_mine.st_size >= bytes)) { close(fd); fd = -1; }  // not sized yet
      if (fd >= 0) {
        p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) return VPIN_ECOMM;
        s = static_cast<CommSeg*>(p);
        auto still_named = [&]() {  // does the name still lead to the object mapped here?
          const int fd2 = shm_open(name, O_RDWR, 0600);
          struct stat st_now;
          const bool y = fd2 >= 0 && fstat(fd2, &st_now) == 0 && st_now.st_ino == st_mine.st_ino && st_now.st_dev == st_mine.st_dev;
          if (fd2 >= 0) close(fd2);
          return y;
        };
        bool ready = false, same = true;
        long polls = 0;
        while (!(ready = s->magic.load(std::memory_order_acquire) == kCommMagic)) {
          if ((++polls & 1023) == 0 && !(same = still_named())) break;  // replaced while we waited: a leftover
          if (!sp.step()) break;
        }
        if (ready) {
          const double age = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count() - s->created_unix_s;
          // older than any rank would wait, no longer named, or created by a process that no longer exists (the segment of a job
          // that died moments ago, which rank 0 of THIS job has not replaced yet: it passes every other check -- ADVICE r4):
          // a leftover.  (kill(pid, 0): ESRCH = gone; EPERM = exists under another user.)
          // The pid test only holds when both processes live in the SAME PID namespace (one container per rank
          // sharing /dev/shm -- --ipc=host, a shared emptyDir -- sees ESRCH for a live creator, and every rank would reject the
          // live segment until the timeout): the creator's namespace inode travels in the segment; in another namespace, or
          // when either side cannot tell, the age and the name decide alone, as they did before the pid test existed.
          const uint64_t my_ns = pid_namespace_id();
          const bool same_ns = my_ns != 0 && s->creator_pidns != 0 && my_ns == s->creator_pidns;
          const bool creator_alive = !same_ns || (s->creator_pid > 0 && (kill((pid_t)s->creator_pid, 0) == 0 || errno == EPERM));
          same = age <= timeout + 5.0 && creator_alive && still_named();
        }
        if (ready && same) break;
        munmap(p, bytes);
        p = MAP_FAILED;
        s = nullptr;
      }
      if (!sp.step()) return VPIN_ECOMM;
      usleep(200);
    }
    if (s->world != (uint32_t)world || s->slot_bytes != slot_bytes) { munmap(p, bytes); return VPIN_ECOMM; }
  }
  vpin_comm* cm = new (std::nothrow) vpin_comm();
  if (!cm) { munmap(p, bytes); return VPIN_ENOMEM; }
  cm->rank = rank; cm->world = world; cm->kind = 0; cm->seg = s; cm->seg_bytes = bytes; cm->timeout_s = timeout;
  snprintf(cm->shm_name, sizeof cm->shm_name, "%s", name);
  s->attached.fetch_add(1);
  // everyone attached: the name can go (the mappings stay), so a crash later leaves nothing behind
  Spinner sp(timeout);
  while (s->attached.load() < (uint32_t)world)
    if (!sp.step()) { s->abort_flag.store(1); if (rank == 0) (void)shm_unlink(name); munmap(p, bytes); delete cm; return VPIN_ECOMM; }
  if (rank == 0) (void)shm_unlink(name);
  cm->t_last_exit = now_s();
  *out = cm;
  return VPIN_OK;
}

int vpin_comm_create_local(int world, size_t slot_bytes, vpin_comm** out) {
  if (!out || world < 1) return VPIN_EINVAL;
  if (slot_bytes == 0) slot_bytes = (size_t)1 << 20;
  const size_t bytes = seg_size(world, slot_bytes);
  void* p = nullptr;
  if (posix_memalign(&p, 4096, bytes) != 0) return VPIN_ENOMEM;
  memset(p, 0, 4096);
  CommSeg* s = static_cast<CommSeg*>(p);
  seg_init(s, world, slot_bytes);
  auto* refs = new std::atomic<int>(world);
  for (int r = 0; r < world; r++) {
    vpin_comm* cm = new vpin_comm();
    cm->rank = r; cm->world = world; cm->kind = 1; cm->seg = s; cm->seg_bytes = bytes; cm->seg_refs = refs;
    cm->timeout_s = env_timeout();
    cm->t_last_exit = now_s();
    out[r] = cm;
  }
  s->attached.store((uint32_t)world);
  return VPIN_OK;
}

int vpin_comm_create_callbacks(int rank, int world, vpin_allgather_fn fn, void* user, vpin_comm** out) {
  if (!out || !fn || world < 1 || rank < 0 || rank >= world) return VPIN_EINVAL;
  vpin_comm* cm = new (std::nothrow) vpin_comm();
  if (!cm) return VPIN_ENOMEM;
  cm->rank = rank; cm->world = world; cm->kind = 2; cm->cb = fn; cm->cb_user = user;
  cm->timeout_s = env_timeout();
  cm->t_last_exit = now_s();
  *out = cm;
  return VPIN_OK;
}

void vpin_comm_destroy(vpin_comm* cm) {
  if (!cm) return;
  if (cm->nccl) {
    Rccl* r = rccl();
    if (r) (void)r->CommDestroy((ncclComm_t)cm->nccl);
  }
  if (cm->h_stage) (void)hipHostFree(cm->h_stage);
  if (cm->seg) {
    token_release(cm);
    if (cm->kind == 0) {
      munmap(cm->seg, cm->seg_bytes);
    } else if (cm->kind == 1 && cm->seg_refs) {
      if (cm->seg_refs->fetch_sub(1) == 1) { free(cm->seg); delete cm->seg_refs; }
    }
  }
  delete cm;
}

void vpin_comm_abort(vpin_comm* cm) { comm_abort(cm); }

int vpin_comm_rank(const vpin_comm* cm) { return cm ? cm->rank : -1; }
int vpin_comm_world(const vpin_comm* cm) { return cm ? cm->world : 0; }

int vpin_comm_allgather(vpin_comm* cm, const void* send, void* recv, size_t bytes) { return comm_allgather(cm, send, recv, bytes); }

int vpin_comm_allgather_dev(vpin_comm* cm, vpin_ctx* c, const void* d_send, void* d_recv, size_t bytes) {
  return comm_allgather_dev(cm, c, d_send, d_recv, bytes);
}

int vpin_comm_enable_rccl(vpin_comm* cm, vpin_ctx* c) {
  if (!cm || !c) return VPIN_EINVAL;
  Rccl* r = rccl();
  // every rank reports whether it can, so that nobody waits inside ncclCommInitRank for a rank that cannot
  std::vector<uint8_t> can((size_t)cm->world), mine(1, r ? 1 : 0);
  int rc = comm_allgather(cm, mine.data(), can.data(), 1);
  if (rc) return rc;
  for (uint8_t b : can) if (!b) return VPIN_ENODEV;
  ncclUniqueId id;
  memset(&id, 0, sizeof id);
  if (cm->rank == 0 && r->GetUniqueId(&id) != ncclSuccess) memset(&id, 0xff, sizeof id);
  std::vector<ncclUniqueId> ids((size_t)cm->world);
  if ((rc = comm_allgather(cm, &id, ids.data(), sizeof id))) return rc;
  bool bad = true;
  for (size_t i = 0; i < sizeof id; i++) bad = bad && ((const uint8_t*)&ids[0])[i] == 0xff;
  if (bad) return VPIN_ECOMM;
  (void)hipSetDevice(c->device);
  ncclComm_t nc = nullptr;
  ncclResult_t e = r->CommInitRank(&nc, cm->world, ids[0], cm->rank);
  uint8_t ok = e == ncclSuccess ? 1 : 0;
  if (!ok) set_last_error(r->GetErrorString ? r->GetErrorString(e) : "ncclCommInitRank", hipErrorUnknown);
  if ((rc = comm_allgather(cm, &ok, can.data(), 1))) return rc;
  bool all = true;
  for (uint8_t b : can) all = all && b;
  if (!all) {
    if (ok) (void)r->CommDestroy(nc);
    return VPIN_ECOMM;
  }
  cm->nccl = nc;
  cm->nccl_ctx = c;
  return VPIN_OK;
}

int vpin_comm_set_serialize(vpin_comm* cm, int on) {
  if (!cm) return VPIN_EINVAL;
  if (cm->world == 1 || cm->kind == 2) return on ? VPIN_EINVAL : VPIN_OK;
  if (on && !cm->serialize) {
    cm->serialize = true;
    return token_acquire(cm);
  }
  if (!on && cm->serialize) {
    token_release(cm);
    cm->serialize = false;
  }
  return VPIN_OK;
}

int vpin_comm_stats_read(vpin_comm* cm, vpin_comm_stats* out, int reset) {
  if (!cm || !out) return VPIN_EINVAL;
  *out = cm->st;
  if (reset) { cm->st = vpin_comm_stats{}; cm->tags.clear(); cm->t_last_exit = now_s(); }
  return VPIN_OK;
}

// one line per call-site tag: "<tag> <collectives> <busy_s> <crit_s>\n"; returns the length needed (incl. the terminator)
size_t vpin_comm_stats_tags(vpin_comm* cm, char* buf, size_t cap) {
  if (!cm) return 0;
  std::string out;
  char line[256];
  for (auto& kv : cm->tags) {
    snprintf(line, sizeof line, "%s %llu %.9f %.9f\n", kv.first.c_str(), (unsigned long long)kv.second.n, kv.second.busy_s, kv.second.crit_s);
    out += line;
  }
  if (buf && cap) {
    const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return out.size() + 1;
}

// `iters` back-to-back all-gathers of `bytes` per rank inside the library (no per-call binding overhead): seconds per
// collective on this rank.  Collective.
int vpin_comm_latency(vpin_comm* cm, size_t bytes, int iters, double* seconds_per_collective) {
  if (!cm || !seconds_per_collective || iters < 1 || bytes > ((size_t)1 << 20)) return VPIN_EINVAL;
  std::vector<uint8_t> sb(bytes ? bytes : 1, 0x5a), rb((bytes ? bytes : 1) * (size_t)cm->world);
  int rc = VPIN_OK;
  for (int i = 0; i < 64 && !rc; i++) rc = comm_allgather(cm, sb.data(), rb.data(), bytes);
  const double t0 = now_s();
  for (int i = 0; i < iters && !rc; i++) rc = comm_allgather(cm, sb.data(), rb.data(), bytes);
  *seconds_per_collective = (now_s() - t0) / iters;
  return rc;
}

int vpin_ctx_set_comm(vpin_ctx* c, vpin_comm* cm) {
  if (!c) return VPIN_EINVAL;
  c->comm = cm;
  c->comm_pub.store(cm, std::memory_order_release);
  return VPIN_OK;
}

}  // extern "C"
