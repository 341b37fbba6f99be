// runtime.hip -- error reporting, device query and HIP-graph capture helpers.
#include "odin_device.h"
#include "odin_internal.h"
#include <cstdint>

static thread_local char g_err[512] = "";

int odin_fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s (code %d)", msg, code);
  return code;
}

static thread_local const char* g_last_path = "";

int odin_check_launch(const char* what) {
  g_last_path = what;
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

// CRC-32C (Castagnoli), slicing-by-8, host side: the checksum of TensorFlow's checkpoint
// (TensorBundle) and event-file formats (odin_ai_amd/tf_checkpoint.py).
static uint32_t g_crc_tab[8][256];
static bool g_crc_ready = false;
static void crc_init() {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
    g_crc_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t)
      g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 0xFF];
  g_crc_ready = true;
}
extern "C" uint32_t odin_crc32c(uint32_t crc, const void* data, size_t n) {
  if (!g_crc_ready) crc_init();
  const unsigned char* p = (const unsigned char*)data;
  uint32_t c = ~crc;
  while (n >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = g_crc_tab[7][lo & 0xFF] ^ g_crc_tab[6][(lo >> 8) & 0xFF] ^ g_crc_tab[5][(lo >> 16) & 0xFF] ^
        g_crc_tab[4][lo >> 24] ^ g_crc_tab[3][hi & 0xFF] ^ g_crc_tab[2][(hi >> 8) & 0xFF] ^
        g_crc_tab[1][(hi >> 16) & 0xFF] ^ g_crc_tab[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = (c >> 8) ^ g_crc_tab[0][(c ^ *p++) & 0xFF];
  return ~c;
}

extern "C" int odin_version(void) { return 102; }

// diagnostics: name of the kernel family the calling thread launched last; a "(bf16x3)" suffix marks
// fp32 work carried through the bf16 matrix pipe (priced separately by bench.py's roofline_split)
extern "C" const char* odin_debug_last_path(void) { return g_last_path; }
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
