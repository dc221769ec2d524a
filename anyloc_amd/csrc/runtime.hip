// Error reporting, launch checking and the per-kernel HIP-event profiler.
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace anyloc {

namespace {
thread_local char g_err[1024] = "";

struct ProfEntry {
  std::string name;
  hipEvent_t start, stop;
  double flops, bytes;
};
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfEntry> g_prof;   // one entry per bracketed launch since the last reset
}  // namespace

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return ANYLOC_ERR_HIP;
}

bool profiling_enabled() { return g_prof_on; }

int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, what);
  return ANYLOC_OK;
}

ProfScope::ProfScope(const char* name, hipStream_t s, double flops, double bytes) : slot(-1), stream(s) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfEntry e;
  e.name = name;
  e.flops = flops;
  e.bytes = bytes;
  if (hipEventCreate(&e.start) != hipSuccess || hipEventCreate(&e.stop) != hipSuccess) return;
  (void)hipEventRecord(e.start, s);
  g_prof.push_back(e);
  slot = (int)g_prof.size() - 1;
}

ProfScope::~ProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  (void)hipEventRecord(g_prof[slot].stop, stream);
}

}  // namespace anyloc

using namespace anyloc;

extern "C" {

int anyloc_version(void) { return ANYLOC_ABI_VERSION; }
const char* anyloc_last_error(void) { return g_err; }

int anyloc_profile_enable(int enable) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = enable != 0;
  return ANYLOC_OK;
}

int anyloc_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& e : g_prof) {
    (void)hipEventDestroy(e.start);
    (void)hipEventDestroy(e.stop);
  }
  g_prof.clear();
  return ANYLOC_OK;
}

int anyloc_profile_dump(char* buf, size_t cap) {
  if (!buf || cap < 3) {
    set_error("anyloc_profile_dump: buffer too small");
    return ANYLOC_ERR_INVALID_ARG;
  }
  struct Agg { long calls = 0; double ms = 0, flops = 0, bytes = 0; };
  std::map<std::string, Agg> agg;
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& e : g_prof) {
      if (hipEventSynchronize(e.stop) != hipSuccess) continue;
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, e.start, e.stop) != hipSuccess) continue;
      Agg& a = agg[e.name];
      a.calls += 1;
      a.ms += ms;
      a.flops += e.flops;
      a.bytes += e.bytes;
    }
  }
  std::string s = "{";
  bool first = true;
  for (auto& kv : agg) {
    char tmp[512];
    snprintf(tmp, sizeof(tmp), "%s\"%s\": {\"calls\": %ld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
             first ? "" : ", ", kv.first.c_str(), kv.second.calls, kv.second.ms, kv.second.flops, kv.second.bytes);
    s += tmp;
    first = false;
  }
  s += "}";
  if (s.size() + 1 > cap) {
    set_error("anyloc_profile_dump: need %zu bytes", s.size() + 1);
    return ANYLOC_ERR_INVALID_ARG;
  }
  memcpy(buf, s.c_str(), s.size() + 1);
  return ANYLOC_OK;
}

}  // extern "C"
