// profile.hip -- kernel timeline of libdram_hip.so (measurement only): hipEvent pairs around every kernel
// launch, recorded on the launch stream, tagged with family / executed MFMA FLOPs / algorithmic HBM bytes.
// bench.py reads the log for its roofline table; off by default.
#include <mutex>
#include <vector>

#include "common.h"

bool g_dram_prof_on = false;

namespace {
struct Rec {
  int family, variant;
  double flops, bytes, alg;
  hipEvent_t a, b;
  bool closed;
};
std::mutex g_mu;                 // forward runs on the caller's thread, backward on autograd's
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;  // events are created once per dram_profile_start and reused
size_t g_pool_used = 0;
int g_max = 0, g_dropped = 0;
// index of the record this THREAD has begun and not yet ended (launches do not nest within a thread, but the
// forward / data-loader thread and autograd's backward thread may interleave theirs), and the timeline
// generation it belongs to (a record index must not survive a dram_profile_start)
thread_local int t_open = -1;
thread_local unsigned t_gen = 0;
unsigned g_gen = 0;

const char* const kNames[DRAM_FAM_COUNT] = {
    "conv_wino2d", "wino_in", "wino_gemm_nn", "wino_out", "wino_gemm_tn", "wino_wgrad_out", "weight_pack",
    "conv_wgrad_w2d", "conv_igemm", "conv_wgrad", "stem", "bn_elementwise", "pool_up", "head_loss", "optim", "prep",
    "conv_bf16", "wgrad_bf16"};
const int kMfma[DRAM_FAM_COUNT] = {1, 0, 1, 0, 1, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1, 1};
}  // namespace

void dram_prof_begin(int family, int variant, double mfma_flops, double hbm_bytes, double alg_flops, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  t_open = -1;
  if ((int)g_recs.size() >= g_max || g_pool_used + 2 > g_pool.size()) { ++g_dropped; return; }
  Rec r{family, variant, mfma_flops, hbm_bytes, alg_flops, g_pool[g_pool_used], g_pool[g_pool_used + 1], false};
  g_pool_used += 2;
  if (hipEventRecord(r.a, s) != hipSuccess) { ++g_dropped; return; }
  g_recs.push_back(r);
  t_open = (int)g_recs.size() - 1;
  t_gen = g_gen;
}

void dram_prof_end(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (t_open < 0 || t_gen != g_gen || t_open >= (int)g_recs.size()) { t_open = -1; return; }
  Rec& r = g_recs[t_open];
  r.closed = hipEventRecord(r.b, s) == hipSuccess;
  t_open = -1;
}

extern "C" const char* dram_profile_family_name(int family) {
  return family >= 0 && family < DRAM_FAM_COUNT ? kNames[family] : "?";
}
extern "C" int dram_profile_family_is_mfma(int family) {
  return family >= 0 && family < DRAM_FAM_COUNT ? kMfma[family] : 0;
}

extern "C" int dram_profile_start(int max_records) {
  if (max_records < 1 || max_records > (1 << 22)) return DRAM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(g_mu);

  g_dram_prof_on = false;
  g_recs.clear();
  g_recs.reserve(max_records);
  while (g_pool.size() < 2 * (size_t)max_records) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return DRAM_ERR_WORKSPACE;
    g_pool.push_back(e);
  }
  g_pool_used = 0;
  g_max = max_records;
  g_dropped = 0;
  ++g_gen;
  g_dram_prof_on = true;
  return DRAM_OK;
}

extern "C" int dram_profile_stop(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_dram_prof_on = false;
  return DRAM_OK;
}

extern "C" int dram_profile_dropped(void) { return g_dropped; }

extern "C" int dram_profile_read(DramProfRecord* out, int max_records) {
  if (!out || max_records < 0) return DRAM_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lk(g_mu);
  int n = 0;
  for (const Rec& r : g_recs) {
    if (n >= max_records) break;
    if (!r.closed) continue;
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    out[n].family = r.family;
    out[n].variant = r.variant;
    out[n].mfma_flops = r.flops;
    out[n].alg_flops = r.alg;
    out[n].hbm_bytes = r.bytes;
    out[n].ms = ms;
    out[n].pad_ = 0.f;
    ++n;
  }
  return n;
}
