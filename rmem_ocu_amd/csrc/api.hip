// Error plumbing, version and hipGraph capture helpers of the C ABI (include/rmem.h).
#include "common.h"
#include "../../include/rmem.h"
#include <string.h>
#include <stdio.h>
#include <atomic>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

extern "C" void rmem_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

int rmem_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  char buf[512];
  snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
  rmem_set_error(buf);
  return -2;
}

extern "C" int rmem_abi_version(void) { return RMEM_ABI_VERSION; }
extern "C" const char* rmem_last_error_string(void) { return g_err; }

#define HIP_TRY(call, what)                  \
  do {                                       \
    const hipError_t e_ = (call);            \
    if (e_ != hipSuccess) {                  \
      char b_[512];                          \
      snprintf(b_, sizeof(b_), "%s: %s", what, hipGetErrorString(e_)); \
      rmem_set_error(b_);                    \
      return -3;                             \
    }                                        \
  } while (0)

extern "C" int rmem_graph_begin(void* stream) {
  HIP_TRY(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal), "rmem_graph_begin");
  return 0;
}

extern "C" int rmem_graph_end(void* stream, void** graph_exec_out) {
  RMEM_REQUIRE(graph_exec_out, "rmem_graph_end: null output");
  hipGraph_t g = nullptr;
  HIP_TRY(hipStreamEndCapture((hipStream_t)stream, &g), "rmem_graph_end: end capture");
  hipGraphExec_t ge = nullptr;
  const hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  HIP_TRY(e, "rmem_graph_end: instantiate");
  *graph_exec_out = (void*)ge;
  return 0;
}

extern "C" int rmem_graph_launch(void* graph_exec, void* stream) {
  HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream), "rmem_graph_launch");
  return 0;
}

extern "C" int rmem_graph_destroy(void* graph_exec) {
  HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)graph_exec), "rmem_graph_destroy");
  return 0;
}

extern "C" int rmem_copy_async(void* dst, const void* src, size_t bytes, void* stream) {
  RMEM_REQUIRE(dst && src && bytes > 0, "rmem_copy_async: bad argument");
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, (hipStream_t)stream), "rmem_copy_async");
  return 0;
}

extern "C" int rmem_copy2d_async(void* dst, long long dst_pitch, const void* src, long long src_pitch, long long row_bytes, int rows,
                                 void* stream) {
  RMEM_REQUIRE(dst && src && row_bytes > 0 && rows > 0 && dst_pitch >= row_bytes && src_pitch >= row_bytes,
               "rmem_copy2d_async: bad argument");
  HIP_TRY(hipMemcpy2DAsync(dst, (size_t)dst_pitch, src, (size_t)src_pitch, (size_t)row_bytes, (size_t)rows, hipMemcpyDeviceToDevice,
                           (hipStream_t)stream), "rmem_copy2d_async");
  return 0;
}

// ---- opt-in launch timers (bench.py's roofline leg): HIP events around selected launches, on the launch stream ----
// Process-wide state, off unless started; `on` is read by every launch without the lock, everything else under `mu`.
namespace {
__global__ void k_prof_nop() {}

struct ProfState {
  std::mutex mu;
  std::atomic<bool> on{false};
  float bracket_ms = 0.f;       // HIP-event bracket cost around an empty kernel (calibrated in start)
  std::vector<hipEvent_t> ev;   // pairs
  std::vector<double> flops;
  size_t used = 0;
  size_t cap = 0;               // launches this start() allows (the event pool may be larger from an earlier start)
};
ProfState g_prof[2];

int prof_start(ProfState& st, int max_launches, const char* who) {
  std::lock_guard<std::mutex> lk(st.mu);
  if (max_launches <= 0) { rmem_set_error("rmem_profile_start: max_launches must be > 0"); return -1; }
  while (st.ev.size() < (size_t)max_launches * 2) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) { rmem_set_error(who); return -3; }
    st.ev.push_back(e);
  }
  st.flops.assign(max_launches, 0.0);
  st.used = 0;
  st.cap = (size_t)max_launches;
  // calibrate what two event records around ONE launch cost by themselves: bracket an empty kernel on an idle stream,
  // keep the minimum of 32 trials; stop() subtracts it from every timed launch
  hipStream_t cs;
  if (hipStreamCreate(&cs) == hipSuccess) {
    float best = 1e9f;
    for (int i = 0; i < 32; ++i) {
      (void)hipEventRecord(st.ev[0], cs);
      hipLaunchKernelGGL(k_prof_nop, dim3(1), dim3(64), 0, cs);
      (void)hipEventRecord(st.ev[1], cs);
      (void)hipEventSynchronize(st.ev[1]);
      float t = 0.f;
      if (hipEventElapsedTime(&t, st.ev[0], st.ev[1]) == hipSuccess && t < best) best = t;
    }
    (void)hipStreamDestroy(cs);
    st.bracket_ms = best < 1e8f ? best : 0.f;
  }
  st.on.store(true);
  return 0;
}

int prof_stop(ProfState& st, double* total_ms, double* total_flops, int* launches, const char* who) {
  std::lock_guard<std::mutex> lk(st.mu);
  st.on.store(false);
  double ms = 0.0, fl = 0.0;
  for (size_t i = 0; i < st.used; ++i) {
    float t = 0.f;
    if (hipEventSynchronize(st.ev[2 * i + 1]) != hipSuccess || hipEventElapsedTime(&t, st.ev[2 * i], st.ev[2 * i + 1]) != hipSuccess) {
      rmem_set_error(who);
      return -3;
    }
    ms += fmaxf(t - st.bracket_ms, 0.f);
    fl += st.flops[i];
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (launches) *launches = (int)st.used;
  return 0;
}
}  // namespace

int rmem_prof_begin(int channel, void* stream, double flops) {
  if (channel < 0 || channel > 1) return -1;
  ProfState& st = g_prof[channel];
  if (!st.on.load(std::memory_order_relaxed)) return -1;
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cs);
  if (cs != hipStreamCaptureStatusNone) return -1;
  int slot = -1;
  {
    std::lock_guard<std::mutex> lk(st.mu);
    if (st.on.load() && st.used < st.cap) { slot = (int)st.used++; st.flops[slot] = flops; }
  }
  if (slot >= 0) (void)hipEventRecord(st.ev[2 * slot], s);
  return slot;
}

void rmem_prof_end(int channel, int slot, void* stream) {
  if (channel >= 0 && channel <= 1 && slot >= 0) (void)hipEventRecord(g_prof[channel].ev[2 * slot + 1], (hipStream_t)stream);
}

extern "C" int rmem_profile_start(int max_launches) { return prof_start(g_prof[RMEM_PROF_MEM_READ], max_launches, "rmem_profile_start: hipEventCreate failed"); }
extern "C" int rmem_profile_stop(double* total_ms, double* total_flops, int* launches) {
  return prof_stop(g_prof[RMEM_PROF_MEM_READ], total_ms, total_flops, launches, "rmem_profile_stop: event query failed");
}
extern "C" int rmem_gated_profile_start(void) { return prof_start(g_prof[RMEM_PROF_GATED_PV], 256, "rmem_gated_profile_start: hipEventCreate failed"); }
extern "C" int rmem_gated_profile_stop(double* total_ms, double* total_flops, int* launches) {
  return prof_stop(g_prof[RMEM_PROF_GATED_PV], total_ms, total_flops, launches, "rmem_gated_profile_stop: event query failed");
}
