// Error reporting, launch checking and the per-kernel HIP-event profiler.
#include <cstdlib>
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
std::string g_prof_filter;       // non-empty: only launches with exactly this tag are bracketed
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

namespace {
struct OptDef {
  const char* name;
  int64_t def;
};
// order = enum Option (common.hpp)
const OptDef kOptDefs[OPT_COUNT] = {
    {"gemm_f32_cfg", 0},      {"x6_cfg", 0},          {"h3_cfg", 0},        {"h3_group_m", 8},   {"h3_tiny_max", 256},
    {"h3_deep_max", 320},     {"h3_deep2_max", 500},  {"h3_epi_lds", 1},    {"ln_rows_per_wave", 0}, {"ln_small_rows", 4096}, {"ln_waves", 8}, {"ln_direct_rows", 1200}, {"h3_fuse", 1},
    {"x6_fuse", 1},           {"h3_min_rows", 0},     {"x6_min_rows", 1600}, {"attn_cfg", 0},    {"attn_x6", -1},
    {"vlad_parts", 0},    {"vlad_two_pass", 0}, {"vlad_fused_v", 0},
    {"kmeans_fused_v", 0},    {"kmeans_max_chunks", 0}, {"h3_mfma16", -1}, {"h3_swiglu_t", 1}, {"h3_fast_silu", 1}, {"topk_fewq_x6", 2}, {"topk_h3", -1},
    {"h3s_cfg", -1}, {"h3s_ksplit", 0}, {"h3s_kb", 0}, {"h3s_stages", 0}, {"h3s_mask", 31}, {"h3s_enable", 1}, {"h3_patch", 1}, {"topk_fewq_qdma", 1},
    {"h3s_w12_tall", 1},      {"attn_h3_qg", 1},     {"attn_h3_ks", 0},
    {"h3s_ln_lead", 0},       {"h3_ln_lead", 0},       {"vlad_gather_v", 0},    {"topk_screen", -1},
};
std::mutex g_opt_mu;
int64_t g_opt[OPT_COUNT];
bool g_opt_init = false;

int find_option(const char* name, size_t len) {
  for (int i = 0; i < OPT_COUNT; ++i)
    if (strlen(kOptDefs[i].name) == len && strncmp(kOptDefs[i].name, name, len) == 0) return i;
  return -1;
}

// defaults, then ANYLOC_OPTIONS="name=value,name=value" (the library's only environment variable; unknown names and
// malformed items are reported once on stderr and ignored).  Caller holds g_opt_mu.
void init_options() {
  if (g_opt_init) return;
  for (int i = 0; i < OPT_COUNT; ++i) g_opt[i] = kOptDefs[i].def;
  g_opt_init = true;
  const char* e = getenv("ANYLOC_OPTIONS");
  while (e && *e) {
    const char* end = strchr(e, ',');
    const size_t len = end ? (size_t)(end - e) : strlen(e);
    const char* eq = static_cast<const char*>(memchr(e, '=', len));
    const int id = eq ? find_option(e, (size_t)(eq - e)) : -1;
    if (id >= 0) g_opt[id] = strtoll(eq + 1, nullptr, 0);
    else if (len) fprintf(stderr, "[anyloc] ANYLOC_OPTIONS: ignoring '%.*s'\n", (int)len, e);
    e = end ? end + 1 : nullptr;
  }
}
}  // namespace

int64_t option(Option o) {
  std::lock_guard<std::mutex> lk(g_opt_mu);
  init_options();
  return g_opt[o];
}

int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, what);
  return ANYLOC_OK;
}

ProfScope::ProfScope(const char* name, hipStream_t s, double flops, double bytes) : slot(-1), stream(s) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof_filter.empty() && g_prof_filter != name) return;
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

int anyloc_set_option(const char* name, int64_t value) {
  ANYLOC_CHECK_ARG(name, "set_option: null name");
  std::lock_guard<std::mutex> lk(g_opt_mu);
  init_options();
  const int id = find_option(name, strlen(name));
  ANYLOC_CHECK_ARG(id >= 0, "set_option: unknown option '%s'", name);
  g_opt[id] = value;
  return ANYLOC_OK;
}

int anyloc_get_option(const char* name, int64_t* value) {
  ANYLOC_CHECK_ARG(name && value, "get_option: null pointer");
  std::lock_guard<std::mutex> lk(g_opt_mu);
  init_options();
  const int id = find_option(name, strlen(name));
  ANYLOC_CHECK_ARG(id >= 0, "get_option: unknown option '%s'", name);
  *value = g_opt[id];
  return ANYLOC_OK;
}

int anyloc_reset_options(void) {
  std::lock_guard<std::mutex> lk(g_opt_mu);
  g_opt_init = false;
  init_options();
  return ANYLOC_OK;
}

int anyloc_profile_enable(int enable) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = enable != 0;
  return ANYLOC_OK;
}

int anyloc_profile_filter(const char* tag) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_filter = tag ? tag : "";
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
