// Error plumbing + version entry points of the C ABI (include/pidm.h).
#include "pidm_common.h"

#include <cstdarg>
#include <cstdio>

namespace pidm {
static thread_local char g_err[512] = "";
long long g_kernel_enqueues = 0;

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}
}  // namespace pidm

extern "C" int pidm_version(void) { return PIDM_ABI_VERSION; }
extern "C" const char* pidm_last_error(void) { return pidm::g_err; }
extern "C" const char* pidm_backend(void) {
#ifdef PIDM_BACKEND_NAME
  return PIDM_BACKEND_NAME;
#else
  return "hip";
#endif
}

// ---------------------------------------------------------------------------------------------------------
// optional per-kernel-class timing with HIP events on the launch stream (bench.py's roofline figures)
// ---------------------------------------------------------------------------------------------------------
#include <algorithm>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <stdlib.h>
#include <string.h>
extern "C" char** environ;
namespace pidm {
struct ProfRec { hipEvent_t a, b; int cls; double work; std::string label; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::string g_prof_label;
// per-kernel marks (see pidm_common.h): one event per launch, durations are intervals between consecutive events
struct MarkRec { hipEvent_t e; const char* name; int cls; double work; };
bool g_prof_marks = false;
const char* g_prof_name = nullptr;
static std::vector<MarkRec> g_marks;
static hipStream_t g_mark_stream = nullptr;
static bool g_mark_trace = false;
static struct { bool valid; int cls; double work; } g_mark_pending = {false, -1, 0.0};
void prof_mark(const char* what) {
  MarkRec m;
  m.name = g_prof_name ? g_prof_name : what;
  g_prof_name = nullptr;
  if (g_mark_trace) fprintf(stderr, "[pidm launch] %s\n", m.name);     // PIDM_TRACE_LAUNCHES=1: the launch sequence of a pass
  m.cls = g_mark_pending.valid ? g_mark_pending.cls : -1;
  m.work = g_mark_pending.valid ? g_mark_pending.work : 0.0;
  g_mark_pending.valid = false;
  (void)hipEventCreate(&m.e);
  (void)hipEventRecord(m.e, g_mark_stream);
  g_marks.push_back(m);
}
bool prof_enabled() { return g_prof_on || g_prof_marks; }
void prof_set_label(const char* label) { g_prof_label = label ? label : ""; }
void prof_begin_launch(int cls, double work, hipStream_t st) {
  if (g_prof_marks) {            // the launch's own mark (PIDM_CHECK_LAUNCH) carries class and work
    g_mark_pending.valid = true; g_mark_pending.cls = cls; g_mark_pending.work = work;
    return;
  }
  ProfRec r;
  r.cls = cls; r.work = work; r.label = g_prof_label;
  (void)hipEventCreate(&r.a);
  (void)hipEventCreate(&r.b);
  (void)hipEventRecord(r.a, st);
  g_prof.push_back(r);
}
void prof_end_launch(hipStream_t st) {
  if (g_prof_marks) return;
  (void)hipEventRecord(g_prof.back().b, st);
}
void prof_cancel_last() {
  if (g_prof_marks) { g_mark_pending.valid = false; return; }
  if (!g_prof.empty()) {
    (void)hipEventDestroy(g_prof.back().a);
    (void)hipEventDestroy(g_prof.back().b);
    g_prof.pop_back();
  }
}
void prof_reclass_last(int cls) {
  if (g_prof_marks) {
    if (g_mark_pending.valid) g_mark_pending.cls = cls;
    else if (!g_marks.empty()) g_marks.back().cls = cls;
    return;
  }
  if (!g_prof.empty()) g_prof.back().cls = cls;
}

// ---- knobs: one environment read per name and process ------------------------------------------------------------------------
namespace {
// `val` points into g_knob_strings, which only ever grows: a pointer knob() handed out stays valid across pidm_reload_knobs()
// (a launcher on another thread may still be reading it), at the price of a few leaked bytes per CHANGED value
struct KnobEntry { bool set; const std::string* val; };
std::mutex g_knob_mu;
std::map<std::string, KnobEntry>* g_knobs = nullptr;     // leaked on purpose: launchers may run during static destruction
std::deque<std::string>* g_knob_strings = nullptr;
KnobEntry read_env(const char* name) {
  const char* v = getenv(name);
  if (!v) return KnobEntry{false, nullptr};
  if (!g_knob_strings) g_knob_strings = new std::deque<std::string>();
  g_knob_strings->emplace_back(v);
  return KnobEntry{true, &g_knob_strings->back()};
}
uint64_t g_knob_sig = 0;
bool g_knob_sig_valid = false;
}  // namespace
const char* knob(const char* name) {
  std::lock_guard<std::mutex> lk(g_knob_mu);
  if (!g_knobs) g_knobs = new std::map<std::string, KnobEntry>();
  auto it = g_knobs->find(name);
  if (it == g_knobs->end()) {
    it = g_knobs->emplace(name, read_env(name)).first;
  }
  return it->second.set ? it->second.val->c_str() : nullptr;
}
uint64_t knob_signature() {
  std::lock_guard<std::mutex> lk(g_knob_mu);
  if (!g_knob_sig_valid) {
    uint64_t h = 1469598103934665603ull;
    for (char** e = ::environ; e && *e; ++e) {
      if (strncmp(*e, "PIDM_", 5) != 0) continue;
      for (const char* c = *e; *c; ++c) {
        h ^= (unsigned char)*c;
        h *= 1099511628211ull;
      }
      h ^= h >> 31;
    }
    g_knob_sig = h;
    g_knob_sig_valid = true;
  }
  return g_knob_sig;
}
}  // namespace pidm

extern "C" int pidm_reload_knobs(void) {
  std::lock_guard<std::mutex> lk(pidm::g_knob_mu);
  // entries are re-read in place, never erased: see KnobEntry
  if (pidm::g_knobs)
    for (auto& kv : *pidm::g_knobs) {
      const char* v = getenv(kv.first.c_str());
      const bool same = kv.second.set ? (v && *kv.second.val == v) : (v == nullptr);
      if (!same) kv.second = pidm::read_env(kv.first.c_str());
    }
  pidm::g_knob_sig_valid = false;
  return 0;
}

extern "C" int pidm_prof_kernels_begin(void* stream) {
  using namespace pidm;
  if (g_prof_marks) return fail("pidm_prof_kernels_begin: already on");
  g_mark_stream = as_stream(stream);
  g_mark_trace = knob("PIDM_TRACE_LAUNCHES") != nullptr;
  g_marks.clear();
  g_mark_pending.valid = false;
  g_prof_name = nullptr;
  MarkRec m{nullptr, "(begin)", -1, 0.0};
  if (hipEventCreate(&m.e) != hipSuccess || hipEventRecord(m.e, g_mark_stream) != hipSuccess) return fail("pidm_prof_kernels_begin: no event");
  g_marks.push_back(m);
  g_prof_marks = true;
  return 0;
}
// Ends the per-kernel timing and writes one line per kernel name, largest total first:
//   name \t launches \t total_ms \t work \t class \n     (work = what the launcher declared: conv FLOPs; 0 = not declared; class -1 = none)
// Returns the number of bytes the table needs (incl. the terminating 0; truncated to cap when larger), or -1.
extern "C" long long pidm_prof_kernels_collect(char* buf, size_t cap) {
  using namespace pidm;
  if (!g_prof_marks) return fail("pidm_prof_kernels_collect: not on");
  g_prof_marks = false;
  struct Agg { double ms = 0, work = 0; long n = 0; int cls = -1; };
  std::map<std::string, Agg> by;
  if (!g_marks.empty()) (void)hipEventSynchronize(g_marks.back().e);
  for (size_t k = 1; k < g_marks.size(); ++k) {
    float t = 0.f;
    (void)hipEventElapsedTime(&t, g_marks[k - 1].e, g_marks[k].e);
    Agg& a = by[g_marks[k].name];
    a.ms += t; a.work += g_marks[k].work; a.n += 1;
    if (g_marks[k].cls >= 0) a.cls = g_marks[k].cls;
  }
  for (auto& m : g_marks) (void)hipEventDestroy(m.e);
  g_marks.clear();
  std::vector<std::pair<std::string, Agg>> v(by.begin(), by.end());
  std::sort(v.begin(), v.end(), [](const auto& x, const auto& y) { return x.second.ms > y.second.ms; });
  std::string out;
  char line[320];
  for (auto& kv : v) {
    snprintf(line, sizeof(line), "%s\t%ld\t%.6f\t%.6e\t%d\n", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.work, kv.second.cls);
    out += line;
  }
  if (buf && cap) {
    const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return (long long)out.size() + 1;
}

extern "C" int pidm_prof_enable(int on) {
  pidm::g_prof_on = on != 0;
  return 0;
}
// sums since the last collect: ms[c], launches[c], work[c] (flops or bytes as declared by the launcher), c < 4
extern "C" int pidm_prof_collect(double* ms, long long* launches, double* work) {
  for (int c = 0; c < 4; ++c) { ms[c] = 0.0; launches[c] = 0; work[c] = 0.0; }
  struct Agg { double ms = 0, work = 0; long n = 0; int cls = 0; };
  std::map<std::string, Agg> by_label;
  const char* dump = pidm::knob("PIDM_PROF_DUMP");   // per-shape table of the timed launches (tools/, A/B measurements)
  for (auto& r : pidm::g_prof) {
    (void)hipEventSynchronize(r.b);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, r.a, r.b);
    if (r.cls >= 0 && r.cls < 4) { ms[r.cls] += t; launches[r.cls] += 1; work[r.cls] += r.work; }
    if (dump) { Agg& a = by_label[r.label]; a.ms += t; a.work += r.work; a.n += 1; a.cls = r.cls; }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  pidm::g_prof.clear();
  if (dump && !by_label.empty()) {
    if (FILE* f = fopen(dump, "a")) {
      fprintf(f, "# class launches total_ms avg_us TFLOP/s label\n");
      for (auto& kv : by_label)
        fprintf(f, "%d %ld %.3f %.1f %.1f %s\n", kv.second.cls, kv.second.n, kv.second.ms, kv.second.ms * 1e3 / kv.second.n,
                kv.second.work / (kv.second.ms * 1e-3) / 1e12, kv.first.c_str());
      fclose(f);
    }
  }
  return 0;
}
