// runtime.hip -- error reporting, device query and HIP-graph capture helpers.
#include "odin_device.h"
#include "odin_internal.h"

static thread_local char g_err[512] = "";

int odin_fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s (code %d)", msg, code);
  return code;
}

int odin_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return -100 - (int)e;
  }
  return 0;
}

int odin_num_cus() {
#ifdef ODIN_SIM
  return 16;
#else
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
#endif
}

extern "C" int odin_version(void) { return 100; }
extern "C" const char* odin_last_error(void) { return g_err; }

#ifdef ODIN_SIM
extern "C" int odin_graph_begin(void*) { return odin_fail(-3, "graphs unavailable in sim"); }
extern "C" int odin_graph_end(void*, void**) { return odin_fail(-3, "graphs unavailable in sim"); }
extern "C" int odin_graph_launch(void*, void*) { return odin_fail(-3, "graphs unavailable in sim"); }
extern "C" int odin_graph_destroy(void*) { return 0; }
#else
extern "C" int odin_graph_begin(void* stream) {
  hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
  return e == hipSuccess ? 0 : odin_fail(-100 - (int)e, hipGetErrorString(e));
}
extern "C" int odin_graph_end(void* stream, void** graph_exec_out) {
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
  if (e != hipSuccess) return odin_fail(-100 - (int)e, hipGetErrorString(e));
  hipGraphExec_t ge = nullptr;
  e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return odin_fail(-100 - (int)e, hipGetErrorString(e));
  *graph_exec_out = (void*)ge;
  return 0;
}
extern "C" int odin_graph_launch(void* graph_exec, void* stream) {
  hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
  return e == hipSuccess ? 0 : odin_fail(-100 - (int)e, hipGetErrorString(e));
}
extern "C" int odin_graph_destroy(void* graph_exec) {
  if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
  return 0;
}
#endif
