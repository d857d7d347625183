// Error plumbing, version and hipGraph capture helpers of the C ABI (include/rmem.h).
#include "common.h"
#include "../../include/rmem.h"
#include <string.h>
#include <stdio.h>

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
