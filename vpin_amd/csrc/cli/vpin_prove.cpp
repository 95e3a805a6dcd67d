// vpin_prove -- command-line prover with the reference binary's contract.
//   reference: vPIN_proof_generation/src/main.rs:14-46 (`cargo run -- <label>`), witness files read by
//   load_data.rs:5-63 and load_data_add.rs:5-103 from the cwd-relative directory rust_files/<label>/.
// Usage: vpin_prove <label> [--seed <hex>] [--device N] [--write-proof <dir>] [--sat-only]
// Randomness: like the reference, which builds a fresh RandomTape from OsRng inside every
// proof_point_add / proof_point_mult (Spartan/src/random.rs:14-20, proof_point_mult.rs:44,
// commit_test.rs:74), every proof gets its own 128 bytes (commit seed | proof seed) from the OS.
// With --seed the per-proof seeds are SHAKE256(master || "vPIN/point_add") and
// SHAKE256(master || "vPIN/point_mult") (master = the hex bytes repeated cyclically to 128): reproducible
// runs, still independent tapes -- equal tapes would make the two proofs share Hyrax row blinds.
// Stdout follows the reference line for line (network / gadget banners / counts / proof size / times /
// totals block).  The proof is the whole SNARK (my_lib_prove) and is verified in-process (my_lib_verify) like
// the reference does; --sat-only proves the R1CS satisfiability proof alone (said on stderr);
// --write-proof dumps the proof with its commitments.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/vpin_hip.h"
#include "../host/transcript.h"

namespace {

[[noreturn]] void die(const std::string& msg) {
  fprintf(stderr, "%s\n", msg.c_str());
  exit(101);  // the reference panics (exit code 101) on every load/parse failure
}

std::string slurp(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) die("Failed to open file: " + path);  // load_data.rs:14 expect("Failed to open file")
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

// minimal JSON: arrays of integers, arrays of arrays of integers, arrays of strings
struct Parser {
  const std::string& s;
  size_t i = 0;
  explicit Parser(const std::string& str) : s(str) {}
  void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) i++; }
  bool eat(char c) { ws(); if (i < s.size() && s[i] == c) { i++; return true; } return false; }
  void need(char c) { if (!eat(c)) die("Failed to parse JSON"); }
  long long integer() {
    ws();
    size_t j = i;
    if (j < s.size() && (s[j] == '-' || s[j] == '+')) j++;
    while (j < s.size() && isdigit((unsigned char)s[j])) j++;
    if (j == i) die("Failed to parse JSON");
    long long v = atoll(s.substr(i, j - i).c_str());
    i = j;
    // tolerate "12.0"
    if (i < s.size() && s[i] == '.') { i++; while (i < s.size() && isdigit((unsigned char)s[i])) i++; }
    return v;
  }
  std::string str() {
    need('"');
    size_t j = s.find('"', i);
    if (j == std::string::npos) die("Failed to parse JSON");
    std::string r = s.substr(i, j - i);
    i = j + 1;
    return r;
  }
};

// N x 32 byte matrix (point_*_byte.json): rows shorter than 32 are zero-extended like the reference's
// fixed [u8;32] fill (point_mult.rs:355-361)
std::vector<uint8_t> load_bytes32(const std::string& path, size_t* n_out) {
  std::string txt = slurp(path);
  Parser p(txt);
  std::vector<uint8_t> out;
  p.need('[');
  size_t n = 0;
  if (!p.eat(']')) {
    do {
      p.need('[');
      uint8_t row[32] = {0};
      size_t k = 0;
      if (!p.eat(']')) {
        do {
          long long v = p.integer();
          if (k < 32) row[k] = (uint8_t)v;
          k++;
        } while (p.eat(','));
        p.need(']');
      }
      out.insert(out.end(), row, row + 32);
      n++;
    } while (p.eat(','));
    p.need(']');
  }
  *n_out = n;
  return out;
}

std::vector<uint8_t> load_flags(const std::string& path) {
  std::string txt = slurp(path);
  Parser p(txt);
  std::vector<uint8_t> out;
  p.need('[');
  if (!p.eat(']')) {
    do out.push_back(p.integer() != 0); while (p.eat(','));
    p.need(']');
  }
  return out;
}

// weight.json: decimal strings parsed as u128 (load_data.rs:18-23)
std::vector<uint8_t> load_weights(const std::string& path, size_t* n_out) {
  std::string txt = slurp(path);
  Parser p(txt);
  std::vector<uint8_t> out;
  size_t n = 0;
  p.need('[');
  if (!p.eat(']')) {
    do {
      std::string d = p.str();
      unsigned __int128 v = 0;
      if (d.empty()) die("Failed to parse weight");
      for (char ch : d) {
        if (!isdigit((unsigned char)ch)) die("Failed to parse weight");
        unsigned __int128 nv = v * 10 + (unsigned)(ch - '0');
        if (nv / 10 != v) die("Failed to parse weight");  // u128 overflow
        v = nv;
      }
      uint8_t b[16];
      memcpy(b, &v, 16);
      out.insert(out.end(), b, b + 16);
      n++;
    } while (p.eat(','));
    p.need(']');
  }
  *n_out = n;
  return out;
}

using Clock = std::chrono::steady_clock;
long long ms_since(Clock::time_point t0) {
  return std::chrono::duration_cast<std::chrono::milliseconds>(Clock::now() - t0).count();
}

struct Result { size_t size = 0; long long gen_ms = 0, ver_ms = 0; };

// VPIN_CLI_TRACE=1: where the reference's "Proof generation time" span goes, on stderr
struct Lap {
  Clock::time_point t = Clock::now();
  const bool on = getenv("VPIN_CLI_TRACE") != nullptr;
  void operator()(const char* what) {
    if (!on) return;
    auto n = Clock::now();
    fprintf(stderr, "[vpin_prove] %-24s %9.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};

void check(int rc, const char* what) {
  if (rc != 0) die(std::string(what) + ": " + vpin_strerror(rc) + " [" + vpin_last_error() + "]");
}

bool g_sat_only = false;

Result prove(vpin_ctx* ctx, vpin_instance* inst, const uint8_t seeds[128], const std::string& dump_prefix, Clock::time_point t0) {
  Lap lap;
  if (vpin_instance_is_sat(inst) != 1) die("assertion failed: instance is not satisfied");  // point_mult.rs:650-651
  lap("is_sat");
  const vpin_r1cs* r = vpin_instance_r1cs(inst);
  size_t cap = g_sat_only ? vpin_sat_proof_max_bytes(r->num_cons, r->num_vars) : vpin_snark_proof_max_bytes(r), len = 0;
  size_t ell = 0;
  while (((size_t)1 << ell) < r->num_vars) ell++;
  size_t L = (size_t)1 << (ell / 2);
  std::vector<uint8_t> proof(cap), cp(32 * L), ci(32 * L), ev(96), comm(g_sat_only ? 0 : vpin_spark_comm_bytes(r));
  size_t comm_len = 0;
  if (g_sat_only)
    check(vpin_sat_prove(ctx, r, vpin_instance_vars_para(inst), vpin_instance_vars_input(inst), vpin_instance_vars(inst),
                         vpin_instance_inputs(inst), seeds, seeds + 64, proof.data(), cap, &len, cp.data(), ci.data(), ev.data(),
                         nullptr, nullptr),
          "vpin_sat_prove");
  else  // SNARK::encode + my_lib_prove (proof_point_mult.rs:38-94): what the reference's "Proof size" measures
    check(vpin_snark_prove(ctx, r, vpin_instance_vars_para(inst), vpin_instance_vars_input(inst), vpin_instance_vars(inst),
                           vpin_instance_inputs(inst), seeds, seeds + 64, proof.data(), cap, &len, comm.data(), comm.size(),
                           &comm_len, cp.data(), ci.data()),
          "vpin_snark_prove");
  lap("prove call");
  Result res;
  res.size = len;
  printf("Proof size: %zu bytes\n", len);
  res.gen_ms = ms_since(t0);
  printf("Proof generation time: %lld ms\n", res.gen_ms);
  {
    // proof_point_mult.rs:103-111: verify in-process, assert, report
    auto t1 = Clock::now();
    int vrc = g_sat_only ? vpin_sat_verify(ctx, proof.data(), len, r->num_cons, r->num_vars, vpin_instance_inputs(inst), r->num_inputs,
                                           ev.data(), cp.data(), ci.data())
                         : vpin_snark_verify(ctx, proof.data(), len, comm.data(), comm_len, vpin_instance_inputs(inst), r->num_inputs,
                                             cp.data(), ci.data());
    if (vrc != 0) die(std::string("assertion failed: proof verification: ") + vpin_strerror(vrc));
    printf("Proof verification successful!\n");
    res.ver_ms = ms_since(t1);
    printf("Proof verification time: %lld ms\n", res.ver_ms);
  }
  if (!dump_prefix.empty()) {
    std::ofstream(dump_prefix + ".proof", std::ios::binary).write((const char*)proof.data(), (std::streamsize)len);
    std::ofstream(dump_prefix + ".comm_para", std::ios::binary).write((const char*)cp.data(), (std::streamsize)cp.size());
    std::ofstream(dump_prefix + ".comm_input", std::ios::binary).write((const char*)ci.data(), (std::streamsize)ci.size());
    if (g_sat_only) std::ofstream(dump_prefix + ".inst_evals", std::ios::binary).write((const char*)ev.data(), 96);
    else std::ofstream(dump_prefix + ".comm", std::ios::binary).write((const char*)comm.data(), (std::streamsize)comm_len);
  }
  return res;
}


// the same span with the instance built on the device (vpin_gadget_point_*_dev): default path
Result prove_dev(vpin_ctx* ctx, vpin_dev_instance* inst, const uint8_t seeds[128], const std::string& dump_prefix, Clock::time_point t0) {
  Lap lap;
  if (vpin_dev_instance_is_sat(ctx, inst) != 1) die("assertion failed: instance is not satisfied");  // point_mult.rs:650-651
  lap("is_sat (device)");
  const vpin_r1cs_dev* r = vpin_dev_instance_r1cs(inst);
  size_t num_cons = 0, num_vars = 0, num_inputs = 0;
  vpin_r1cs_dims(r, &num_cons, &num_vars, &num_inputs);
  size_t cap = g_sat_only ? vpin_sat_proof_max_bytes(num_cons, num_vars) : vpin_dev_instance_proof_max_bytes(inst), len = 0;
  size_t ell = 0;
  while (((size_t)1 << ell) < num_vars) ell++;
  size_t L = (size_t)1 << (ell / 2);
  std::vector<uint8_t> proof(cap), cp(32 * L), ci(32 * L), ev(96), comm(g_sat_only ? 0 : vpin_dev_instance_comm_bytes(inst));
  size_t comm_len = 0;
  if (g_sat_only)
    check(vpin_sat_prove_resident(ctx, r, vpin_dev_instance_vars_para(inst), vpin_dev_instance_vars_input(inst),
                                  vpin_dev_instance_vars(inst), vpin_dev_instance_inputs(inst), seeds, seeds + 64, proof.data(), cap,
                                  &len, cp.data(), ci.data(), ev.data(), nullptr, nullptr),
          "vpin_sat_prove_resident");
  else
    check(vpin_snark_prove_dev(ctx, inst, seeds, seeds + 64, proof.data(), cap, &len, comm.data(), comm.size(), &comm_len, cp.data(),
                               ci.data()),
          "vpin_snark_prove_dev");
  lap("prove call");
  Result res;
  res.size = len;
  printf("Proof size: %zu bytes\n", len);
  res.gen_ms = ms_since(t0);
  printf("Proof generation time: %lld ms\n", res.gen_ms);
  {
    auto t1 = Clock::now();
    int vrc = g_sat_only ? vpin_sat_verify(ctx, proof.data(), len, num_cons, num_vars, vpin_dev_instance_inputs(inst), num_inputs,
                                           ev.data(), cp.data(), ci.data())
                         : vpin_snark_verify(ctx, proof.data(), len, comm.data(), comm_len, vpin_dev_instance_inputs(inst), num_inputs,
                                             cp.data(), ci.data());
    if (vrc != 0) die(std::string("assertion failed: proof verification: ") + vpin_strerror(vrc));
    printf("Proof verification successful!\n");
    res.ver_ms = ms_since(t1);
    printf("Proof verification time: %lld ms\n", res.ver_ms);
  }
  if (!dump_prefix.empty()) {
    std::ofstream(dump_prefix + ".proof", std::ios::binary).write((const char*)proof.data(), (std::streamsize)len);
    std::ofstream(dump_prefix + ".comm_para", std::ios::binary).write((const char*)cp.data(), (std::streamsize)cp.size());
    std::ofstream(dump_prefix + ".comm_input", std::ios::binary).write((const char*)ci.data(), (std::streamsize)ci.size());
    if (g_sat_only) std::ofstream(dump_prefix + ".inst_evals", std::ios::binary).write((const char*)ev.data(), 96);
    else std::ofstream(dump_prefix + ".comm", std::ios::binary).write((const char*)comm.data(), (std::streamsize)comm_len);
  }
  return res;
}

// The generator sets are functions of the instance sizes alone, and every set of a label is a prefix of one stream: the
// window table of the LONGEST stream of the run serves both instances.  The run's second instance (point multiplication)
// is the larger one, so its sets are prepared first, inside the first instance's span -- otherwise the first proof builds
// tables that the second one has to build again, longer.  Best effort: a weight file that cannot be read is met again,
// and reported, where the reference would meet it.
size_t count_strings(const std::string& path) {  // entries of weight.json (an array of decimal strings), 0 when unreadable
  std::ifstream f(path, std::ios::binary);
  if (!f) return 0;
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string txt = ss.str();
  size_t q = 0;
  for (char ch : txt) q += ch == '"';
  return q / 2;
}

void prepare_for_mult(vpin_ctx* ctx, const std::string& weight_path, bool sat_only) {
  const size_t nw = count_strings(weight_path);
  size_t nc = 0, nv = 0, nnz[3] = {0, 0, 0};
  if (!nw || vpin_gadget_shape(1, nw, &nc, &nv, nnz) != 0) return;
  const size_t mx = nnz[0] > nnz[1] ? (nnz[0] > nnz[2] ? nnz[0] : nnz[2]) : (nnz[1] > nnz[2] ? nnz[1] : nnz[2]);
  Lap lap;
  lap("  shape of the mult instance");
  if (!sat_only) (void)vpin_spark_prepare(ctx, nc, nv, mx);
  lap("  vpin_spark_prepare");
  (void)vpin_sat_prepare(ctx, nv);
  lap("  vpin_sat_prepare");
}

// number of operations a label's point-mult instance will have (0: none / no file)
size_t mult_ops_of(const std::string& network) {
  if (network == "L2" || network == "L4") return 0;
  return count_strings("rust_files/" + network + "/pointMult/weight.json");
}

struct Opts {
  std::string dump_dir;
  int device = 0;
  bool have_seed = false, host_gadgets = false, no_prefetch = false;
  uint8_t seeds[128];
};

// One label = what one `cargo run -- <label>` of the reference does (main.rs:14-46): the point-add SNARK, the point-mult
// SNARK, the totals block.  *pctx is created inside the first label's point-add span (as a process per label would) and
// kept for the labels after it; `prefetch_largest` names the label whose point-mult shape the generator sets are prepared
// for before the first proof (the largest of the run: every smaller set is a prefix of it).
void run_label(const std::string& network, const Opts& o, vpin_ctx** pctx, const std::string& prefetch_label, size_t n_labels) {
  // one independent 128-byte seed per proof (a shared seed gives both SNARKs the same blinds)
  uint8_t seeds_add[128], seeds_mult[128];
  if (!o.have_seed) {  // OsRng (Spartan/src/random.rs:17), drawn per proof
    std::random_device rd;
    for (auto& b : seeds_add) b = (uint8_t)rd();
    for (auto& b : seeds_mult) b = (uint8_t)rd();
  } else {
    auto derive = [&](const char* domain, uint8_t out[128]) {
      vpin_host::Shake256 sh;
      sh.absorb(o.seeds, 128);
      sh.absorb(reinterpret_cast<const uint8_t*>(domain), strlen(domain));
      sh.finalize();
      sh.squeeze(out, 128);
    };
    derive("vPIN/point_add", seeds_add);
    derive("vPIN/point_mult", seeds_mult);
  }
  printf("network: %s\n", network.c_str());
  vpin_ctx*& ctx = *pctx;
  const std::string base = "rust_files/" + network + "/";

  // ---- point addition (proof_point_add.rs) ----
  auto t0 = Clock::now();
  size_t n1 = 0, n2 = 0, n3 = 0, n4 = 0;
  std::vector<uint8_t> apx = load_bytes32(base + "pointAdd/point_add_px_byte.json", &n1);
  std::vector<uint8_t> apy = load_bytes32(base + "pointAdd/point_add_py_byte.json", &n2);
  std::vector<uint8_t> arx = load_bytes32(base + "pointAdd/point_add_rx_byte.json", &n3);
  std::vector<uint8_t> ary = load_bytes32(base + "pointAdd/point_add_ry_byte.json", &n4);
  std::vector<uint8_t> arz = load_flags(base + "pointAdd/point_add_rz_byte.json");
  if (n2 != n1 || n3 != n1 || n4 != n1 || arz.size() != n1) die("point addition witness files disagree on the number of operations");
  printf("Point Addition Gadget...\n");
  printf("Number of Point Additions: %zu\n", n1);
  Lap lap;
  if (!ctx) {
    const char* pe = getenv("VPIN_CLI_MAIN_PRIO");
    check(vpin_ctx_create_prio(o.device, pe ? atoi(pe) : 0, &ctx), "vpin_ctx_create");
    lap("ctx_create");
    // this process proves n_labels times per instance shape at most and exits: generator tables sized for that
    // (VPIN_CLI_FULL_TABLES: the service's)
    if (!getenv("VPIN_CLI_FULL_TABLES")) check(vpin_ctx_set_expected_proofs(ctx, (int)n_labels), "vpin_ctx_set_expected_proofs");
    if (!prefetch_label.empty() && !o.no_prefetch) {
      prepare_for_mult(ctx, "rust_files/" + prefetch_label + "/pointMult/weight.json", g_sat_only);
      lap("generator sets (mult size)");
    }
  }
  Result ra;
  const std::string add_prefix = o.dump_dir.empty() ? "" : o.dump_dir + "/" + network + "_add";
  if (o.host_gadgets || n1 == 0) {
    vpin_instance* add = nullptr;
    check(vpin_gadget_point_add(apx.data(), apy.data(), arx.data(), ary.data(), arz.data(), n1, &add), "vpin_gadget_point_add");
    lap("gadget_point_add");
    ra = prove(ctx, add, seeds_add, add_prefix, t0);
    vpin_instance_free(add);
  } else {
    vpin_dev_instance* add = nullptr;
    check(vpin_gadget_point_add_dev(ctx, apx.data(), apy.data(), arx.data(), ary.data(), arz.data(), n1, &add), "vpin_gadget_point_add_dev");
    lap("gadget_point_add (device)");
    ra = prove_dev(ctx, add, seeds_add, add_prefix, t0);
    vpin_dev_instance_free(ctx, add);
  }
  printf("\n");

  // ---- point multiplication (proof_point_mult.rs); L2/L4 have none (main.rs:24-31) ----
  Result rm;
  if (network == "L2" || network == "L4") {
    printf("Number of Point Multiplications: 0\n");
    printf("Proof size: 0 bytes\n");
    printf("Proof generation time: 0 ms\n");
    printf("Proof verification time: 0 ms\n");
  } else {
    t0 = Clock::now();
    lap("(between gadgets)");
    size_t nw = 0, m1 = 0, m2 = 0;
    std::vector<uint8_t> w = load_weights(base + "pointMult/weight.json", &nw);
    std::vector<uint8_t> mpx = load_bytes32(base + "pointMult/point_mult_px_byte.json", &m1);
    std::vector<uint8_t> mpy = load_bytes32(base + "pointMult/point_mult_py_byte.json", &m2);
    if (m1 != nw || m2 != nw) die("point multiplication witness files disagree on the number of operations");
    printf("Point Multiplication Gadget...\n");
    printf("Number of Point Multiplications: %zu\n", nw);
    printf("Generating Proof...\n");
    lap("load mult json");
    const std::string mult_prefix = o.dump_dir.empty() ? "" : o.dump_dir + "/" + network + "_mult";
    if (o.host_gadgets || nw == 0) {
      vpin_instance* mult = nullptr;
      check(vpin_gadget_point_mult(w.data(), mpx.data(), mpy.data(), nw, &mult), "vpin_gadget_point_mult");
      lap("gadget_point_mult");
      printf("Still working on...\n");
      rm = prove(ctx, mult, seeds_mult, mult_prefix, t0);
      vpin_instance_free(mult);
    } else {
      vpin_dev_instance* mult = nullptr;
      check(vpin_gadget_point_mult_dev(ctx, w.data(), mpx.data(), mpy.data(), nw, &mult), "vpin_gadget_point_mult_dev");
      lap("gadget_point_mult (device)");
      printf("Still working on...\n");
      rm = prove_dev(ctx, mult, seeds_mult, mult_prefix, t0);
      vpin_dev_instance_free(ctx, mult);
    }
  }
  printf("\n====================================\n");
  printf("Total proof size: %zu bytes\n", ra.size + rm.size);
  printf("Total proof generation time: %lld ms\n", ra.gen_ms + rm.gen_ms);
  printf("Total proof verification time: %lld ms\n", ra.ver_ms + rm.ver_ms);
  printf("====================================\n");
  fflush(stdout);
}

}  // namespace

// vpin_prove <label> [<label> ...] [options]
// One label: the reference binary (`cargo run -- <label>`, main.rs:14-46).  Several labels: what script.sh:205-211 does with
// a process per label (L1 .. L7 of a LeNet trace), in ONE process -- the HIP context, the generator derivations, the host
// fixed-base tables and the device window tables are built once, for the largest instance of the run, and every label
// prints the reference's stdout block unchanged (with --seed the proofs are the ones the one-label runs give).
int main(int argc, char** argv) {
  std::vector<std::string> labels;
  Opts o;
  o.no_prefetch = getenv("VPIN_CLI_NO_PREFETCH") != nullptr;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    if (a == "--device" && i + 1 < argc) o.device = atoi(argv[++i]);
    else if (a == "--write-proof" && i + 1 < argc) o.dump_dir = argv[++i];
    else if (a == "--sat-only") g_sat_only = true;
    else if (a == "--no-prefetch") o.no_prefetch = true;  // every instance builds its own generator sets, in its own span
    else if (a == "--host-gadgets") o.host_gadgets = true;  // build instance + witness on the host cores, upload, then prove
    else if (a == "--seed" && i + 1 < argc) {
      std::string h = argv[++i];  // hex, repeated cyclically to 128 bytes: commit seed | proof seed
      size_t usable = h.size() & ~(size_t)1;
      if (usable == 0) die("--seed needs hex bytes");
      for (size_t k = 0; k < 128; k++) o.seeds[k] = (uint8_t)strtol(h.substr((2 * k) % usable, 2).c_str(), nullptr, 16);
      o.have_seed = true;
    } else if (a.rfind("--", 0) == 0) die(("unknown option " + a).c_str());
    else labels.push_back(a);
  }
  if (labels.empty()) labels.push_back("1");  // main.rs:16
  fprintf(stderr, g_sat_only ? "vpin_prove: R1CS satisfiability proof only (--sat-only)\n"
                             : "vpin_prove: whole SNARK (sat proof + SPARK evaluation proof)\n");
  // the label with the most point multiplications: its generator sets are prepared first
  std::string largest;
  size_t most = 0;
  for (auto& l : labels) {
    const size_t n = mult_ops_of(l);
    if (n > most) { most = n; largest = l; }
  }
  vpin_ctx* ctx = nullptr;
  for (size_t k = 0; k < labels.size(); k++) {
    if (k) printf("\n");
    run_label(labels[k], o, &ctx, largest, labels.size());
  }
  // hand every block back explicitly: VRAM released by hipFree is wiped by the driver in the background, VRAM
  // reclaimed at process teardown is wiped when the next process allocates it (tools/ubench_malloc*.hip)
  vpin_ctx_destroy(ctx);
  vpin_gens_shared_clear();
  return 0;
}
