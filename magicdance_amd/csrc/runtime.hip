// Runtime pieces of libmagicdance_hip.so: ABI version, HIP-graph capture of a launch sequence (one DDIM step =
// ~2.5k kernel launches replayed from a single hipGraphLaunch), and per-kernel-family HIP-event timing used by
// bench.py's roofline leg.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "md_common.h"

namespace md {

int g_last_hip_error = 0;

namespace {
struct ProfRec {
  int family;
  hipEvent_t start, stop;
  double flops, bytes;
  char tag[128];
};
bool g_prof_on = false;
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof_recs;
}  // namespace

ProfScope::ProfScope(int fam, hipStream_t s, double flops, double bytes, const char* tag)
    : family(fam), stream(s), start(nullptr), active(false) {
  if (!g_prof_on) return;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return;
  ProfRec r;
  r.family = fam;
  r.flops = flops;
  r.bytes = bytes;
  r.tag[0] = 0;
  if (tag) {
    strncpy(r.tag, tag, sizeof(r.tag) - 1);
    r.tag[sizeof(r.tag) - 1] = 0;
  }
  if (hipEventCreate(&r.start) != hipSuccess) return;
  if (hipEventCreate(&r.stop) != hipSuccess) {
    (void)hipEventDestroy(r.start);
    return;
  }
  (void)hipEventRecord(r.start, s);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_recs.push_back(r);
  start = r.start;
  active = true;
}

ProfScope::~ProfScope() {
  if (!active) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto it = g_prof_recs.rbegin(); it != g_prof_recs.rend(); ++it) {
    if (it->start == start) {
      (void)hipEventRecord(it->stop, stream);
      break;
    }
  }
}

}  // namespace md

extern "C" int md_version(void) { return 10; }
extern "C" int md_last_hip_error(void) { return md::g_last_hip_error; }
extern "C" const char* md_arch(void) { return "gfx950"; }

extern "C" int md_graph_begin(void* stream) {
  MD_HIP_CHECK(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return MD_OK;
}

extern "C" int md_graph_end(void* stream, void** graph_exec) {
  if (!graph_exec) return MD_ERR_BAD_ARG;
  hipGraph_t graph = nullptr;
  MD_HIP_CHECK(hipStreamEndCapture((hipStream_t)stream, &graph));
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return md::hip_fail(e);
  *graph_exec = (void*)exec;
  return MD_OK;
}

extern "C" int md_graph_launch(void* graph_exec, void* stream) {
  if (!graph_exec) return MD_ERR_BAD_ARG;
  MD_HIP_CHECK(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return MD_OK;
}

extern "C" int md_graph_destroy(void* graph_exec) {
  if (!graph_exec) return MD_OK;
  MD_HIP_CHECK(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return MD_OK;
}

extern "C" int md_prof_enable(int32_t on) {
  md::g_prof_on = on != 0;
  return MD_OK;
}

extern "C" int md_prof_collect(double* ms, int64_t* launches, double* flops, double* bytes) {
  std::lock_guard<std::mutex> lk(md::g_prof_mu);
  for (int i = 0; i < MD_FAM_COUNT; ++i) {
    if (ms) ms[i] = 0.0;
    if (launches) launches[i] = 0;
    if (flops) flops[i] = 0.0;
    if (bytes) bytes[i] = 0.0;
  }
  int rc = MD_OK;
  // MD_PROF_DUMP=<path>: append one line per launch (family, ms, flops, bytes, shape tag) for offline tuning
  const char* dump_path = getenv("MD_PROF_DUMP");
  FILE* dump = dump_path ? fopen(dump_path, "a") : nullptr;
  for (auto& r : md::g_prof_recs) {
    float t = 0.f;
    hipError_t e = hipEventSynchronize(r.stop);
    if (e == hipSuccess) e = hipEventElapsedTime(&t, r.start, r.stop);
    if (e != hipSuccess) {
      rc = md::hip_fail(e);
    } else if (r.family >= 0 && r.family < MD_FAM_COUNT) {
      if (ms) ms[r.family] += (double)t;
      if (launches) launches[r.family] += 1;
      if (flops) flops[r.family] += r.flops;
      if (bytes) bytes[r.family] += r.bytes;
      if (dump) fprintf(dump, "%d\t%.6f\t%.0f\t%.0f\t%s\n", r.family, (double)t, r.flops, r.bytes, r.tag);
    }
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  if (dump) fclose(dump);
  md::g_prof_recs.clear();
  return rc;
}
