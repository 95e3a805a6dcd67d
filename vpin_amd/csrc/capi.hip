// capi.hip -- context, table management and profiling behind include/vpin_hip.h
#include <chrono>
#include <csignal>
#include <execinfo.h>
#include <fcntl.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "ctx.h"

namespace vpin {

static thread_local std::string g_last_error;

void set_last_error(const char* what, hipError_t e) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
}

static hipEvent_t get_event(vpin_ctx* c) {
  if (!c->free_events.empty()) {
    hipEvent_t e = c->free_events.back();
    c->free_events.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

// live contexts of this process (vpin_ctx_create .. vpin_ctx_destroy): lets a handle freed without a context find
// out whether its owner still exists, and lets an allocation that fails reclaim the blocks other contexts cache
static std::mutex g_ctx_mu;
static std::vector<vpin_ctx*> g_live_ctxs;
// a context another thread is working on (reclaiming its cached blocks, returning a table to its pool) is pinned: its
// destroy waits until the pins are gone
static void ctx_wait_unpinned(vpin_ctx* c) {
  while (c->pins.load(std::memory_order_acquire) != 0) std::this_thread::yield();
}

void ctx_register(vpin_ctx* c) { std::lock_guard<std::mutex> g(g_ctx_mu); g_live_ctxs.push_back(c); }
void ctx_unregister(vpin_ctx* c) {
  std::lock_guard<std::mutex> g(g_ctx_mu);
  for (size_t i = 0; i < g_live_ctxs.size(); i++)
    if (g_live_ctxs[i] == c) { g_live_ctxs.erase(g_live_ctxs.begin() + (long)i); break; }
}
int live_ctx_count() {
  std::lock_guard<std::mutex> g(g_ctx_mu);
  return (int)g_live_ctxs.size();
}
// CPUs this process may use at once: its cgroup's CFS quota (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us) when there is
// one, else the affinity mask.  A one-GPU box hands a job 16 CPUs of a 256-thread host: omp_get_num_procs() says 256, and a
// process that runs more than 16 threads for a while is stopped for the rest of every 100 ms period (round 6: the ~20 ms stalls of
// the W = 8 rehearsal -- 8 rank-threads x teams of 4).
double host_cpu_quota() {
  static const double q = [] {
    double v = 0.0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char a[32] = {0};
      long per = 0;
      if (fscanf(f, "%31s %ld", a, &per) == 2 && strcmp(a, "max") != 0 && per > 0) v = atof(a) / (double)per;
      fclose(f);
    } else if (FILE* f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
      long quota = -1, per = 100000;
      if (fscanf(f1, "%ld", &quota) != 1) quota = -1;
      fclose(f1);
      if (FILE* f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f2, "%ld", &per) != 1) per = 100000; fclose(f2); }
      if (quota > 0 && per > 0) v = (double)quota / (double)per;
    }
    const double hw = (double)std::thread::hardware_concurrency();
    return v > 0.0 && (hw <= 0.0 || v < hw) ? v : (hw > 0.0 ? hw : 1.0);
  }();
  return q;
}
bool ctx_is_live(vpin_ctx* c) {
  std::lock_guard<std::mutex> g(g_ctx_mu);
  for (auto* x : g_live_ctxs) if (x == c) return true;
  return false;
}

// VPIN_POOL_TRACE=<MiB>: every allocation / release of at least that size on stderr with the bytes in use afterwards and the
// milliseconds since the first traced event; hipMalloc / hipFree calls with what they took (a multi-GB hipMalloc or the hipFree
// of a context's cached blocks is the usual suspect when a span jumps by hundreds of ms)
static size_t pool_trace_min() {
  static const size_t v = [] { const char* e = getenv("VPIN_POOL_TRACE"); return e ? ((size_t)atol(e) << 20) : (size_t)0; }();
  return v;
}
static double pool_trace_ms() {
  static const auto t0 = std::chrono::steady_clock::now();
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
static std::atomic<long long> g_in_use{0};
// calls into the driver's allocator by this library, process-wide (vpin_driver_alloc_stats): a proof of a shape its context has
// already proven must make none -- a multi-GB hipMalloc costs 0.3 ms most of the time and 0.2-2.8 s every few calls
std::atomic<unsigned long long> g_driver_allocs{0}, g_driver_alloc_bytes{0};
void note_driver_alloc(size_t bytes) { g_driver_allocs.fetch_add(1, std::memory_order_relaxed); g_driver_alloc_bytes.fetch_add(bytes, std::memory_order_relaxed); }
static void pool_trace(const char* what, size_t cls, bool fresh, double took_ms = 0.0) {
  if (!pool_trace_min() || cls < pool_trace_min()) return;
  fprintf(stderr, "[pool %10.2f ms] %-7s %9.1f MiB%s  in use %9.1f MiB", pool_trace_ms(), what, (double)cls / 1048576.0,
          fresh ? " (hipMalloc)" : "", (double)g_in_use.load() / 1048576.0);
  if (took_ms > 0.0) fprintf(stderr, "  took %.2f ms", took_ms);
  fputc('\n', stderr);
}

// cached (free-listed) blocks of one context back to the driver.  The free lists are taken out under the pool lock; the
// stream sync (work that last used the blocks is ordered on that stream) and the hipFree calls happen with no lock held, so
// another lane's in-flight proof does not stall every context of the process behind a global mutex.
static void pool_release_unlocked(vpin_ctx* c) {
  std::vector<void*> blocks;
  {
    std::lock_guard<std::mutex> g(c->pool_mu);
    for (auto& kv : c->pool_free_lists)
      for (void* p : kv.second) { c->pool_sizes.erase(p); blocks.push_back(p); }
    c->pool_free_lists.clear();
  }
  if (blocks.empty()) return;
  const double t_a = pool_trace_min() ? pool_trace_ms() : 0.0;
  // both streams of the context: a proof that has switched to its CU-masked second stream (vpin_ctx_set_cumask_after_phase1) may
  // have the blocks' last use in flight on either, and another thread's reclaim must not read c->stream while it is being
  // switched -- stream_main / stream_alt only change in calls that own the context
  if (c->stream_main) (void)hipStreamSynchronize(c->stream_main);
  if (c->stream_alt) (void)hipStreamSynchronize(c->stream_alt);
  if (!c->stream_main) (void)hipStreamSynchronize(c->stream);
  for (void* p : blocks) (void)hipFree(p);
  if (pool_trace_min())
    fprintf(stderr, "[pool %10.2f ms] release %zu cached blocks of ctx %p to the driver: %.2f ms\n", pool_trace_ms(), blocks.size(), (void*)c,
            pool_trace_ms() - t_a);
}

// hipMalloc only when the device reports room for the block (+ 256 MiB): round 6 saw an out-of-memory hipMalloc take the process
// down INSIDE the HIP runtime (a 4 GiB request of a split 2^25 proof on a nearly full device: libamdhip64 -> libhsa-runtime64 ->
// pthread_mutex_lock, SIGSEGV) instead of returning hipErrorOutOfMemory.  Blocks below 64 MiB are not asked about (the query costs
// more than they do, and they are not what runs a device out of memory).  A failed check is reported exactly like a failed hipMalloc.
hipError_t driver_malloc(void** p, size_t bytes) {
  if (bytes >= ((size_t)64 << 20)) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < bytes + ((size_t)256 << 20)) return hipErrorOutOfMemory;
  }
  return hipMalloc(p, bytes);
}

int dev_alloc(vpin_ctx* c, size_t bytes, void** out) {
  size_t cls = 256;
  while (cls < bytes) cls <<= 1;
  if (cls > (size_t)1 << 22) cls = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);  // MiB granularity above 4 MiB
  {
    std::lock_guard<std::mutex> g(c->pool_mu);
    // exact class first; for large requests any cached block up to 2x the size will do (proofs of
    // different instance sizes would otherwise each leave their own multi-GB blocks in the pool)
    for (auto it = c->pool_free_lists.lower_bound(cls); it != c->pool_free_lists.end(); ++it) {
      if (it->first != cls && (cls < ((size_t)1 << 22) || it->first > 2 * cls)) break;
      if (it->second.empty()) continue;
      *out = it->second.back();
      it->second.pop_back();
      if (pool_trace_min()) { g_in_use += (long long)it->first; pool_trace("alloc", it->first, false); }
      return VPIN_OK;
    }
  }
  void* p = nullptr;
  const double t_m = pool_trace_min() ? pool_trace_ms() : 0.0;
  if (driver_malloc(&p, cls) != hipSuccess) {
    (void)hipGetLastError();
    if (pool_trace_min()) fprintf(stderr, "[pool %10.2f ms] hipMalloc of %.1f MiB FAILED: releasing cached blocks\n", pool_trace_ms(), (double)cls / 1048576.0);
    dev_pool_release(c);  // give this context's cached blocks back and retry
    if (driver_malloc(&p, cls) != hipSuccess) {
      (void)hipGetLastError();
      // the other contexts on this device (the lanes of a shared GPU) cache multi-GB blocks too
      std::vector<vpin_ctx*> others;
      {
        std::lock_guard<std::mutex> g(g_ctx_mu);  // only to copy and pin the list
        // not the peers of a collective proof, nor a context whose persistent tail kernel is resident: releasing their
        // blocks synchronises their stream, and that stream may be waiting for a reply which depends on THIS rank's next
        // all-gather -- both ranks would sit out the comm timeout instead of recovering memory
        for (auto* x : g_live_ctxs)
          if (x != c && x->device == c->device && x->comm_pub.load(std::memory_order_acquire) == nullptr &&
              x->tail_rounds.load(std::memory_order_acquire) == 0) {
            x->pins.fetch_add(1, std::memory_order_acq_rel);
            others.push_back(x);
          }
      }
      for (auto* x : others) {
        // re-checked after pinning: the owner may have started a tail or joined a collective proof since the list was made
        // (a tail that becomes resident after THIS check only delays the release by its rounds: the owner keeps replying)
        if (x->comm_pub.load(std::memory_order_acquire) == nullptr && x->tail_rounds.load(std::memory_order_acquire) == 0)
          pool_release_unlocked(x);
        x->pins.fetch_sub(1, std::memory_order_acq_rel);
      }
      if (driver_malloc(&p, cls) != hipSuccess) { (void)hipGetLastError(); return VPIN_ENOMEM; }
    }
  }
  note_driver_alloc(cls);
  std::lock_guard<std::mutex> g(c->pool_mu);
  c->pool_sizes[p] = cls;
  *out = p;
  if (pool_trace_min()) { g_in_use += (long long)cls; pool_trace("alloc", cls, true, pool_trace_ms() - t_m); }
  return VPIN_OK;
}

void dev_free(vpin_ctx* c, void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> g(c->pool_mu);
  auto it = c->pool_sizes.find(p);
  if (it == c->pool_sizes.end()) { (void)hipFree(p); return; }
  c->pool_free_lists[it->second].push_back(p);
  if (pool_trace_min()) { g_in_use -= (long long)it->second; pool_trace("free", it->second, false); }
}

void dev_pool_release(vpin_ctx* c) { pool_release_unlocked(c); }

// a pooled block back to the pool of the context that allocated it, whichever context (or none) the caller holds: the owner is
// looked up and pinned under the registry lock, so a concurrent vpin_ctx_destroy of it waits; an owner that is gone has released
// the block in its destroy.  (freeing through another context's dev_free hipFree'd the block and left the owner's
// pool_sizes with a stale entry.)
void dev_free_owned(vpin_ctx* owner, vpin_ctx* fallback, void* p) {
  if (!p) return;
  vpin_ctx* pool = owner ? owner : fallback;
  bool live = false;
  if (pool) {
    std::lock_guard<std::mutex> g(g_ctx_mu);
    for (auto* x : g_live_ctxs) live = live || x == pool;
    if (live) pool->pins.fetch_add(1, std::memory_order_acq_rel);
  }
  if (live) { dev_free(pool, p); pool->pins.fetch_sub(1, std::memory_order_acq_rel); }
  else if (!owner) (void)hipFree(p);
}

void dev_release_block(vpin_ctx* c, void* p) {
  if (!p) return;
  size_t sz = 0;
  {
    std::lock_guard<std::mutex> g(c->pool_mu);
    auto it = c->pool_sizes.find(p);
    if (it != c->pool_sizes.end()) { sz = it->second; c->pool_sizes.erase(it); }
  }
  const double t_a = pool_trace_min() ? pool_trace_ms() : 0.0;
  (void)hipFree(p);
  if (pool_trace_min()) { g_in_use -= (long long)sz; pool_trace("hipFree", sz, false, pool_trace_ms() - t_a); }
}

static void ctx_switch_stream(vpin_ctx* c, hipStream_t to, int cus, bool masked) {
  if (!to || c->stream == to) return;
  (void)hipEventRecord(c->stream_switch_ev, c->stream);
  (void)hipStreamWaitEvent(to, c->stream_switch_ev, 0);
  c->stream = to;
  c->num_cus = cus;
  c->cu_masked = masked;
}
void ctx_enter_alt(vpin_ctx* c) { if (c && c->stream_alt) ctx_switch_stream(c, c->stream_alt, c->cus_alt, true); }
void ctx_leave_alt(vpin_ctx* c) { if (c && c->stream_alt) ctx_switch_stream(c, c->stream_main, c->cus_main, false); }

ProfScope::ProfScope(vpin_ctx* c, int kclass, double bytes, int also, double units) : ctx(c) {
  if (!c->prof) return;
  ProfRec r;
  r.kclass = kclass;
  r.also = also;
  r.units = units;
  r.bytes = bytes;
  r.start = get_event(c);
  r.stop = get_event(c);
  (void)hipEventRecord(r.start, c->stream);
  c->recs.push_back(r);
  rec = (int)c->recs.size() - 1;
}

ProfScope::~ProfScope() {
  if (rec >= 0) (void)hipEventRecord(ctx->recs[rec].stop, ctx->stream);
}

}  // namespace vpin

using namespace vpin;

extern "C" {

const char* vpin_strerror(int code) {
  switch (code) {
    case VPIN_OK: return "ok";
    case VPIN_EINVAL: return "invalid argument";
    case VPIN_ENODEV: return "no usable HIP device (this library has no CPU fallback)";
    case VPIN_ENOMEM: return "out of memory";
    case VPIN_EHIP: return "HIP runtime error";
    case VPIN_ESHAPE: return "operand shapes do not match";
    case VPIN_EVERIFY: return "proof rejected by the verifier";
    case VPIN_ECOMM: return "a peer rank failed or timed out";
    default: return "unknown error";
  }
}

const char* vpin_last_error(void) { return g_last_error.c_str(); }

int vpin_abi_version(void) { return 2; }

static int ctx_create(int device, int priority, const uint32_t* cu_mask, uint32_t n_words, vpin_ctx** out);

int vpin_ctx_create(int device, vpin_ctx** out) { return ctx_create(device, 0, nullptr, 0, out); }
int vpin_ctx_create_prio(int device, int priority, vpin_ctx** out) { return ctx_create(device, priority, nullptr, 0, out); }
int vpin_ctx_create_cumask(int device, const uint32_t* cu_mask, uint32_t n_words, vpin_ctx** out) {
  if (!cu_mask || n_words == 0) return VPIN_EINVAL;
  return ctx_create(device, 0, cu_mask, n_words, out);
}

static int ctx_create(int device, int priority, const uint32_t* cu_mask, uint32_t n_words, vpin_ctx** out) {
  if (!out) return VPIN_EINVAL;
  // host OpenMP teams must wait passively (see prover.cpp host_threads); set before the runtime starts
  setenv("KMP_BLOCKTIME", "0", 0);
  setenv("OMP_WAIT_POLICY", "PASSIVE", 0);
  // the runtime's machine-topology discovery for thread affinity costs 30-40 ms of a cold process on a 256-thread host;
  // the small host teams of this library (<= 8 threads, milliseconds of work) are not pinned anyway
  setenv("KMP_AFFINITY", "disabled", 0);
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return VPIN_ENODEV;
  if (device < 0 || device >= n) return VPIN_EINVAL;
  VPIN_HIP_TRY(hipSetDevice(device));
  vpin_ctx* c = new (std::nothrow) vpin_ctx();
  if (!c) return VPIN_ENOMEM;
  c->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cus = prop.multiProcessorCount;
  // priority: < 0 = scheduled ahead of normal streams (for latency-bound proofs of small instances that
  // share the device with a large one), > 0 = behind them; clamped to the device's range
  int lo = 0, hi = 0;  // numerically: hi <= 0 <= lo
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  int prio = priority < 0 ? hi : priority > 0 ? lo : 0;
  hipError_t e;
  if (cu_mask) {
    // the enabled CUs are what the row-commitment kernels size their grids to (msm.hip: resident strip workgroups)
    int enabled = 0;
    for (uint32_t w = 0; w < n_words; w++) {
      uint32_t bits = cu_mask[w];
      if ((w + 1) * 32 > (uint32_t)c->num_cus) bits &= (w * 32 >= (uint32_t)c->num_cus) ? 0u : ((1u << (c->num_cus - w * 32)) - 1u);
      enabled += __builtin_popcount(bits);
    }
    if (enabled == 0) { delete c; return VPIN_EINVAL; }
    e = hipExtStreamCreateWithCUMask(&c->stream, n_words, cu_mask);
    c->num_cus = enabled;
    c->cu_masked = true;
  } else {
    e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio);
  }
  if (e != hipSuccess) { set_last_error("hipStreamCreate", e); delete c; return VPIN_EHIP; }
  c->stream_main = c->stream;
  c->cus_main = c->num_cus;
  c->partials_cap = (size_t)18 * 4096 * 3;  // kSparkMaxInst x kRoundBlocks x 3 (spark.hip); sumcheck.hip needs kMaxBlocks x 3
  if (hipMalloc(&c->d_partials, c->partials_cap * sizeof(fq)) != hipSuccess ||
      hipMalloc(&c->d_out, 8 * sizeof(fq)) != hipSuccess ||
      hipHostMalloc(&c->h_out, 8 * sizeof(fq), hipHostMallocDefault) != hipSuccess) {
    vpin_ctx_destroy(c);
    return VPIN_ENOMEM;
  }
  ctx_register(c);
  *out = c;
  return VPIN_OK;
}

void vpin_ctx_destroy(vpin_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  ctx_leave_alt(c);
  if (c->prover_cache_free) c->prover_cache_free(c);
  if (c->spark_cache_free) c->spark_cache_free(c);
  if (c->verify_cache_free) c->verify_cache_free(c);
  if (c->h_spark) (void)hipHostFree(c->h_spark);
  if (c->d_spark_cnt) (void)hipFree(c->d_spark_cnt);
  if (c->d_tail_cnt) (void)hipFree(c->d_tail_cnt);
  if (c->d_tail_red) (void)hipFree(c->d_tail_red);
  if (c->d_add_count) (void)hipFree(c->d_add_count);
  ctx_unregister(c);
  ctx_wait_unpinned(c);
  dev_pool_release(c);
  // Handles (tables, device instances, decommitments) must be freed before their context.  Blocks still out are
  // released here so the VRAM is not lost; a handle freed later with this (dead) context would touch freed
  // memory, so say so loudly instead of failing silently.
  if (!c->pool_sizes.empty()) {
    fprintf(stderr, "vpin_ctx_destroy: %zu device block(s) still referenced by live handles are released with the context; "
                    "free tables / instances / decommitments before the context\n", c->pool_sizes.size());
    for (auto& kv : c->pool_sizes) (void)hipFree(kv.first);
    c->pool_sizes.clear();
  }
  for (auto& r : c->recs) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
  for (auto e : c->free_events) (void)hipEventDestroy(e);
  if (c->d_partials) (void)hipFree(c->d_partials);
  if (c->d_out) (void)hipFree(c->d_out);
  if (c->h_out) (void)hipHostFree(c->h_out);
  if (c->h_bullet) (void)hipHostFree(c->h_bullet);
  if (c->stream_alt) { (void)hipStreamSynchronize(c->stream_alt); (void)hipStreamDestroy(c->stream_alt); }
  if (c->stream_switch_ev) (void)hipEventDestroy(c->stream_switch_ev);
  if (c->stream_main) (void)hipStreamDestroy(c->stream_main);
  delete c;
}

void* vpin_ctx_stream(vpin_ctx* c) { return c ? (void*)c->stream : nullptr; }

int vpin_ctx_set_cumask_after_phase1(vpin_ctx* c, const uint32_t* cu_mask, uint32_t n_words) {
  if (!c || c->cu_masked || c->stream != c->stream_main) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  if (c->stream_alt) {  // replace / clear
    VPIN_HIP_TRY(hipStreamSynchronize(c->stream_alt));
    (void)hipStreamDestroy(c->stream_alt);
    c->stream_alt = nullptr;
    c->cus_alt = 0;
  }
  if (!cu_mask || n_words == 0) return VPIN_OK;
  int enabled = 0;
  for (uint32_t w = 0; w < n_words; w++) {
    uint32_t bits = cu_mask[w];
    if ((w + 1) * 32 > (uint32_t)c->cus_main) bits &= (w * 32 >= (uint32_t)c->cus_main) ? 0u : ((1u << (c->cus_main - w * 32)) - 1u);
    enabled += __builtin_popcount(bits);
  }
  if (enabled == 0) return VPIN_EINVAL;
  if (!c->stream_switch_ev) VPIN_HIP_TRY(hipEventCreateWithFlags(&c->stream_switch_ev, hipEventDisableTiming));
  VPIN_HIP_TRY(hipExtStreamCreateWithCUMask(&c->stream_alt, n_words, cu_mask));
  c->cus_alt = enabled;
  return VPIN_OK;
}

int vpin_ctx_set_progress_flag(vpin_ctx* c, int* flag) {
  if (!c) return VPIN_EINVAL;
  c->progress_flag = flag;
  return VPIN_OK;
}

unsigned long long vpin_ctx_strip_rows_taken(vpin_ctx* c) { return c ? c->strip_rows_taken : 0ull; }
unsigned long long vpin_ctx_pip_row_chunks(vpin_ctx* c) { return c ? c->pip_row_chunks : 0ull; }

int vpin_host_register(void* p, size_t bytes) {
  if (!p || !bytes) return VPIN_EINVAL;
  VPIN_HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
  return VPIN_OK;
}
int vpin_host_unregister(void* p) {
  if (!p) return VPIN_EINVAL;
  VPIN_HIP_TRY(hipHostUnregister(p));
  return VPIN_OK;
}

int vpin_ctx_pool_trim(vpin_ctx* c) {
  if (!c) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  dev_pool_release(c);
  return VPIN_OK;
}

int vpin_ctx_pool_stats(vpin_ctx* c, size_t out[3]) {
  if (!c || !out) return VPIN_EINVAL;
  std::lock_guard<std::mutex> g(c->pool_mu);
  size_t total = 0, cached = 0;
  for (auto& kv : c->pool_sizes) total += kv.second;
  for (auto& kv : c->pool_free_lists) cached += kv.first * kv.second.size();
  out[0] = total; out[1] = cached; out[2] = c->pool_sizes.size();
  return VPIN_OK;
}

namespace {
char g_crash_buf[8192];
volatile size_t g_crash_len = 0;
struct sigaction g_crash_old[5];
const int g_crash_sigs[5] = {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGTERM};
void crash_handler(int) {
  const size_t n = g_crash_len;
  size_t off = 0;
  while (off < n) {
    const ssize_t w = write(1, g_crash_buf + off, n - off);
    if (w <= 0) break;
    off += (size_t)w;
  }
  _exit(0);
}
}  // namespace

// VPIN_SEGV_TRACE=1 (development aid): a SIGSEGV / SIGBUS / SIGABRT prints the faulting thread's call stack (module + offset per
// frame: addr2line -e libvpin_hip.so resolves them) before the default action takes the process down
namespace {
int g_segv_fd = 2;  // VPIN_SEGV_TRACE=<path>: the stack goes to that file (a test runner may have captured fd 2); =1: stderr
void segv_trace_handler(int sig) {
  void* frames[64];
  const int n = backtrace(frames, 64);
  const char msg[] = "[vpin] fatal signal, call stack of the faulting thread:\n";
  (void)!write(g_segv_fd, msg, sizeof msg - 1);
  backtrace_symbols_fd(frames, n, g_segv_fd);
  signal(sig, SIG_DFL);
  raise(sig);
}
struct SegvTraceInit {
  SegvTraceInit() {
    const char* e = getenv("VPIN_SEGV_TRACE");
    if (!e) return;
    if (e[0] == '/' || e[0] == '.' || strchr(e, '/')) {
      const int fd = open(e, O_WRONLY | O_CREAT | O_APPEND, 0644);
      if (fd >= 0) g_segv_fd = fd;
    }
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = segv_trace_handler;
    sigemptyset(&sa.sa_mask);
    for (int sg : {SIGSEGV, SIGBUS, SIGABRT}) (void)sigaction(sg, &sa, nullptr);
  }
} g_segv_trace_init;
}  // namespace

int vpin_crash_line_set(const char* line, size_t n) {
  if (n > sizeof g_crash_buf || (n && !line)) return VPIN_EINVAL;
  const bool was_armed = g_crash_len != 0;
  if (n == 0) {
    if (was_armed)
      for (int i = 0; i < 5; i++) (void)sigaction(g_crash_sigs[i], &g_crash_old[i], nullptr);
    g_crash_len = 0;
    return VPIN_OK;
  }
  memcpy(g_crash_buf, line, n);
  g_crash_len = n;
  if (!was_armed) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = crash_handler;
    sigemptyset(&sa.sa_mask);
    for (int i = 0; i < 5; i++) (void)sigaction(g_crash_sigs[i], &sa, &g_crash_old[i]);
  }
  return VPIN_OK;
}

int vpin_driver_alloc_stats(unsigned long long out[2]) {
  if (!out) return VPIN_EINVAL;
  out[0] = vpin::g_driver_allocs.load(std::memory_order_relaxed);
  out[1] = vpin::g_driver_alloc_bytes.load(std::memory_order_relaxed);
  return VPIN_OK;
}

int vpin_ctx_set_shared_device(vpin_ctx* c, int on) {
  if (!c) return VPIN_EINVAL;
  c->shared_device = on != 0;
  return VPIN_OK;
}

int vpin_ctx_set_low_memory(vpin_ctx* c, int on) {
  if (!c) return VPIN_EINVAL;
  c->low_memory = on != 0;
  return VPIN_OK;
}

int vpin_ctx_set_expected_proofs(vpin_ctx* c, int n) {
  if (!c || n < 0) return VPIN_EINVAL;
  c->expected_proofs = n;
  return VPIN_OK;
}

int vpin_ctx_device_props(vpin_ctx* c, int* num_cus, int* clock_khz) {
  if (!c || !num_cus || !clock_khz) return VPIN_EINVAL;
  hipDeviceProp_t prop;
  VPIN_HIP_TRY(hipGetDeviceProperties(&prop, c->device));
  *num_cus = prop.multiProcessorCount;
  *clock_khz = prop.clockRate;
  return VPIN_OK;
}

int vpin_ctx_mem_info(vpin_ctx* c, size_t* free_bytes, size_t* total_bytes) {
  if (!c || !free_bytes || !total_bytes) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  VPIN_HIP_TRY(hipMemGetInfo(free_bytes, total_bytes));
  return VPIN_OK;
}

int vpin_ctx_sync(vpin_ctx* c) {
  if (!c) return VPIN_EINVAL;
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return VPIN_OK;
}

}  // extern "C"

namespace vpin {
// table whose every element the caller is about to overwrite: no zero fill (a 1 GiB memset per
// table per proof is pure HBM traffic)
int table_alloc_uninit(vpin_ctx* c, size_t len, vpin_table** out) {
  if (!c || !out || !is_pow2(len)) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  vpin_table* t = new (std::nothrow) vpin_table();
  if (!t) return VPIN_ENOMEM;
  if (dev_alloc(c, len * sizeof(fq), (void**)&t->d) != VPIN_OK) { delete t; return VPIN_ENOMEM; }
  t->len = t->cap = len;
  t->owner = c;
  *out = t;
  return VPIN_OK;
}
}  // namespace vpin

extern "C" {

int vpin_table_alloc(vpin_ctx* c, size_t len, vpin_table** out) {
  if (!c || !out || !is_pow2(len)) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  vpin_table* t = new (std::nothrow) vpin_table();
  if (!t) return VPIN_ENOMEM;
  if (dev_alloc(c, len * sizeof(fq), (void**)&t->d) != VPIN_OK) { delete t; return VPIN_ENOMEM; }
  t->len = t->cap = len;
  t->owner = c;
  hipError_t e = hipMemsetAsync(t->d, 0, len * sizeof(fq), c->stream);
  if (e != hipSuccess) { set_last_error("hipMemsetAsync", e); dev_free(c, t->d); delete t; return VPIN_EHIP; }
  *out = t;
  return VPIN_OK;
}

int vpin_table_upload(vpin_ctx* c, const uint8_t* mont32, size_t len, vpin_table** out) {
  if (!c || !out || !mont32 || !is_pow2(len)) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  vpin_table* t = new (std::nothrow) vpin_table();
  if (!t) return VPIN_ENOMEM;
  if (dev_alloc(c, len * sizeof(fq), (void**)&t->d) != VPIN_OK) { delete t; return VPIN_ENOMEM; }
  t->len = t->cap = len;
  t->owner = c;
  hipError_t e = hipMemcpyAsync(t->d, mont32, len * sizeof(fq), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) { set_last_error("hipMemcpy H2D", e); dev_free(c, t->d); delete t; return VPIN_EHIP; }
  *out = t;
  return VPIN_OK;
}

int vpin_table_wrap(vpin_ctx* c, void* device_ptr, size_t len, vpin_table** out) {
  if (!c || !out || !device_ptr || !is_pow2(len) || ((uintptr_t)device_ptr & 15)) return VPIN_EINVAL;
  vpin_table* t = new (std::nothrow) vpin_table();
  if (!t) return VPIN_ENOMEM;
  t->d = (fq*)device_ptr;
  t->len = t->cap = len;
  t->owned = false;
  *out = t;
  return VPIN_OK;
}

int vpin_table_clone(vpin_ctx* c, const vpin_table* src, vpin_table** out) {
  if (!c || !src || !out) return VPIN_EINVAL;
  (void)hipSetDevice(c->device);
  vpin_table* t = new (std::nothrow) vpin_table();
  if (!t) return VPIN_ENOMEM;
  if (dev_alloc(c, src->len * sizeof(fq), (void**)&t->d) != VPIN_OK) { delete t; return VPIN_ENOMEM; }
  t->len = t->cap = src->len;
  t->owner = c;
  hipError_t e = hipMemcpyAsync(t->d, src->d, src->len * sizeof(fq), hipMemcpyDeviceToDevice, c->stream);
  if (e != hipSuccess) { set_last_error("hipMemcpy D2D", e); dev_free(c, t->d); delete t; return VPIN_EHIP; }
  *out = t;
  return VPIN_OK;
}

void vpin_table_free(vpin_ctx* c, vpin_table* t) {
  if (!t) return;
  if (t->owned && t->d) {
    // The block goes back to the pool it came from: the table's recorded owner (whatever context the caller passes; a
    // block returned to another context's pool would be released twice).  The owner is looked up and pinned under the
    // registry lock, so a concurrent vpin_ctx_destroy of it waits; if the owner is gone, its destroy released the block.
    vpin_ctx* pool = t->owner ? t->owner : c;
    bool live = false;
    if (pool) {
      std::lock_guard<std::mutex> g(g_ctx_mu);
      for (auto* x : g_live_ctxs) live = live || x == pool;
      if (live) pool->pins.fetch_add(1, std::memory_order_acq_rel);
    }
    if (live) { dev_free(pool, t->d); pool->pins.fetch_sub(1, std::memory_order_acq_rel); }
    else if (!t->owner) (void)hipFree(t->d);
  }
  delete t;
}

size_t vpin_table_len(const vpin_table* t) { return t ? t->len : 0; }
void* vpin_table_device_ptr(const vpin_table* t) { return t ? (void*)t->d : nullptr; }

int vpin_table_read(vpin_ctx* c, const vpin_table* t, size_t off, size_t n, uint8_t* out) {
  if (!c || !t || !out) return VPIN_EINVAL;
  if (off + n > t->len) return VPIN_ESHAPE;
  if (n == 0) return VPIN_OK;
  (void)hipSetDevice(c->device);
  VPIN_HIP_TRY(hipMemcpyAsync(out, t->d + off, n * sizeof(fq), hipMemcpyDeviceToHost, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return VPIN_OK;
}

int vpin_table_write(vpin_ctx* c, vpin_table* t, size_t off, size_t n, const uint8_t* src) {
  if (!c || !t || !src) return VPIN_EINVAL;
  if (off + n > t->cap) return VPIN_ESHAPE;
  if (n == 0) return VPIN_OK;
  (void)hipSetDevice(c->device);
  VPIN_HIP_TRY(hipMemcpyAsync(t->d + off, src, n * sizeof(fq), hipMemcpyHostToDevice, c->stream));
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  return VPIN_OK;
}

int vpin_prof_enable(vpin_ctx* c, int on) {
  if (!c) return VPIN_EINVAL;
  c->prof = on != 0;
  c->prof_count_adds = on >= 2;
  if (c->prof_count_adds && !c->d_add_count) {
    (void)hipSetDevice(c->device);
    VPIN_HIP_TRY(hipMalloc((void**)&c->d_add_count, sizeof(unsigned long long)));
    VPIN_HIP_TRY(hipMemsetAsync(c->d_add_count, 0, sizeof(unsigned long long), c->stream));
    VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  }
  return VPIN_OK;
}

static int prof_drain(vpin_ctx* c) {
  if (c->recs.empty()) return VPIN_OK;
  VPIN_HIP_TRY(hipStreamSynchronize(c->stream));
  for (auto& r : c->recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
      c->stats[r.kclass].launches += 1;
      c->stats[r.kclass].ms += ms;
      c->stats[r.kclass].alg_bytes += r.bytes;
      c->stats[r.kclass].units += r.units;
      if (r.also >= 0 && r.also < VPIN_K_COUNT) {
        c->stats[r.also].launches += 1;
        c->stats[r.also].ms += ms;
        c->stats[r.also].alg_bytes += r.bytes;
        c->stats[r.also].units += r.units;
      }
    }
    c->free_events.push_back(r.start);
    c->free_events.push_back(r.stop);
  }
  c->recs.clear();
  if (c->d_add_count) {  // table additions counted since the last drain (prof level 2)
    unsigned long long n = 0;
    if (hipMemcpyAsync(&n, c->d_add_count, sizeof n, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
        hipStreamSynchronize(c->stream) == hipSuccess && n) {
      c->stats[VPIN_K_MSM_ROWS].units += (double)n;
      (void)hipMemsetAsync(c->d_add_count, 0, sizeof n, c->stream);
    }
  }
  return VPIN_OK;
}

int vpin_prof_reset(vpin_ctx* c) {
  if (!c) return VPIN_EINVAL;
  int rc = prof_drain(c);
  memset(c->stats, 0, sizeof(c->stats));
  return rc;
}

int vpin_prof_read(vpin_ctx* c, vpin_kstat* stats) {
  if (!c || !stats) return VPIN_EINVAL;
  int rc = prof_drain(c);
  memcpy(stats, c->stats, sizeof(c->stats));
  return rc;
}

}  // extern "C"
