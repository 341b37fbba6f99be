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

extern "C" int odin_version(void) { return 103; }

// ---- range words of gradient tensors (odin_device.h: odin_range_shift) -----------------------------------------
namespace {
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ t, size_t n, unsigned* word) {
  // (a tensor that does not start on a 16-byte boundary -- a view into a flat buffer -- is walked from its first
  // aligned element; workgroup 0 takes the unaligned head and the tail)
  const size_t head = (size_t)((4u - (unsigned)(((uintptr_t)t >> 2) & 3u)) & 3u) < n
                          ? (size_t)((4u - (unsigned)(((uintptr_t)t >> 2) & 3u)) & 3u) : n;
  const size_t n4 = (n - head) >> 2;
  const float4* t4 = reinterpret_cast<const float4*>(t + head);
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = t4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    // (fmaxf drops NaNs: a NaN element propagates through the consumer's arithmetic, not through its scale)
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x < head) m = fmaxf(m, fabsf(t[threadIdx.x]));
    const size_t done = head + (n4 << 2);
    if (threadIdx.x < n - done) m = fmaxf(m, fabsf(t[done + threadIdx.x]));
  }
  __shared__ float red[16];
  odin_amax_commit_wg(word, m, threadIdx.x, 256, red, blockIdx.x);
}
// Zeroing is done by KERNELS, never by hipMemsetAsync: a memset node captured into the step graph was not ordered
// against its neighbours on replay (ROCm 7.2: the scratch word of a fallback was cleared AFTER absmax_kernel had
// filled it in about one process out of two -- the consumer then scaled by 2^115 and every product overflowed)
__global__ void range_zero_kernel(unsigned* word) { word[threadIdx.x * ODIN_RANGE_STRIDE] = 0u; }
__global__ __launch_bounds__(256) void zero_u32_kernel(unsigned* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
constexpr int RANGE_SCRATCH_BLOCKS = 64;
#ifdef ODIN_SIM
unsigned g_range_scratch[RANGE_SCRATCH_BLOCKS * ODIN_RANGE_WORDS];
#else
__device__ unsigned g_range_scratch[RANGE_SCRATCH_BLOCKS * ODIN_RANGE_WORDS];
#endif
}  // namespace

extern "C" int odin_range_reset(uint32_t* words, int n, void* stream) {
  if (words == nullptr || n <= 0) return 0;
  return odin_zero_u32(words, (size_t)n * ODIN_RANGE_WORDS, stream);
}

int odin_zero_u32(uint32_t* p, size_t n, void* stream) {
  if (n == 0) return 0;
  size_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  ODIN_LAUNCH(zero_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (unsigned*)p, n);
  return odin_check_launch("zero_u32");
}

extern "C" int odin_absmax(const float* t, size_t n, uint32_t* word, void* stream) {
  if (t == nullptr || word == nullptr) return odin_fail(-2, "odin_absmax: null argument");
  if (n == 0) return 0;
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  ODIN_LAUNCH(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, t, n, (unsigned*)word);
  return odin_check_launch("absmax");
}

#include <atomic>
static std::atomic<int> g_absmax_fallbacks{0};
extern "C" int odin_debug_absmax_fallbacks(void) { return g_absmax_fallbacks.load(); }

// producer side of the range contract (gather_conv.hip: track_dx): a kernel family that does not track its outputs
// is followed by one pass that folds max|t| into the caller's word (atomicMax: the word keeps what it already holds)
int odin_absmax_fold(const float* t, size_t n, uint32_t* word, void* stream) {
  ++g_absmax_fallbacks;
  return odin_absmax(t, n, word, stream);
}

// the word a consumer reads: the caller's, or a library scratch word filled by one pass over the tensor.  The
// scratch words are a ring of 64 per device (a __device__ array has one address PER DEVICE: resolved for the current
// device, not cached from the first), its position one process-wide atomic counter: launches on one stream are
// ordered, and 64 fallbacks later the word's reader has long finished.  Callers that spread consumers over several
// streams hand their own words (the engine always does).
const uint32_t* odin_range_word_of(const float* t, size_t n, const uint32_t* given, void* stream) {
  if (given != nullptr) return given;
  constexpr int MAX_DEV = 16;
  static std::atomic<unsigned*> bases[MAX_DEV] = {};
  static std::atomic<unsigned> next{0};
  int dev = 0;
#ifndef ODIN_SIM
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) { (void)hipGetLastError(); return nullptr; }
#endif
  unsigned* base = bases[dev].load();
  if (base == nullptr) {
#ifdef ODIN_SIM
    base = g_range_scratch;
#else
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_range_scratch)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    base = (unsigned*)q;
#endif
    bases[dev].store(base);
  }
  unsigned* w = base + (size_t)(next.fetch_add(1u) % RANGE_SCRATCH_BLOCKS) * ODIN_RANGE_WORDS;
  ++g_absmax_fallbacks;
  ODIN_LAUNCH(range_zero_kernel, dim3(1), dim3(ODIN_RANGE_SLOTS), 0, stream, w);
  if (odin_absmax(t, n, w, stream) != 0) return nullptr;
  return w;
}

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
