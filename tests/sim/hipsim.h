// hipsim.h -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// A tiny lock-step fiber emulator that lets the *same* .hip kernel sources be compiled
// as host C++ (-DODIN_SIM) and executed on the CPU: one ucontext fiber per HIP thread,
// one workgroup at a time, round-robin scheduling, __syncthreads() as a block barrier,
// and wave64 cross-lane operations (f32 MFMA 32x32x2 / 16x16x4, __shfl_*) emulated by a
// per-wave rendezvous.  Purpose: debug the indexing / tiling logic of the gfx950 kernels
// in the build container (which has no GPU) and run them under host sanitizers.  The
// MFMA emulation follows the operand / accumulator lane maps documented in
// /opt/skills/guides/cdna_hip_programming.md section 3 (A: lane l holds A[l&31][l>>5],
// B: B[l>>5][l&31], C/D: col=lane&31,row=(r&3)+8*(r>>2)+4*(lane>>5)), evaluated as the
// k-ordered fmaf chain the hardware performs.
#pragma once
#include <ucontext.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
struct int4 { int x, y, z, w; };
struct uint4 { unsigned x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return {x, y}; }

namespace sim {

constexpr int WAVE = 64;
constexpr size_t STACK = 256 * 1024;

struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
  dim3 tid;
  int lin = 0;
};

struct WaveXchg {
  float a[2][WAVE], b[2][WAVE];
  unsigned long long arrive = 0;  // total arrivals
};

struct State {
  std::vector<Fiber> fibers;
  ucontext_t sched;
  Fiber* cur = nullptr;
  dim3 blockIdx, blockDim, gridDim;
  unsigned long long barrier_arrive = 0;
  int nthreads = 0;
  std::vector<WaveXchg> waves;
  std::vector<unsigned long long> wave_gen;  // per-fiber generation counter for wave ops
  std::vector<unsigned long long> bar_gen;
  void* dyn_smem = nullptr;
  std::function<void()> body;
};

inline State& S() { static State s; return s; }

inline void yield() {
  State& s = S();
  swapcontext(&s.cur->ctx, &s.sched);
}

inline void block_barrier() {
  State& s = S();
  int me = s.cur->lin;
  unsigned long long gen = ++s.bar_gen[me];
  s.barrier_arrive++;
  // all threads of the block must reach generation `gen`
  while (s.barrier_arrive < gen * (unsigned long long)s.nthreads) yield();
}

// rendezvous of the 64 lanes of the calling wave; returns the exchange parity slot
inline int wave_deposit(float a, float b) {
  State& s = S();
  int me = s.cur->lin, w = me / WAVE, lane = me % WAVE;
  unsigned long long gen = ++s.wave_gen[me];
  int slot = (int)(gen & 1);
  WaveXchg& x = s.waves[w];
  x.a[slot][lane] = a;
  x.b[slot][lane] = b;
  x.arrive++;
  int lanes = std::min(WAVE, s.nthreads - w * WAVE);
  while (x.arrive < gen * (unsigned long long)lanes) yield();
  return slot;
}

typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

inline f32x16_t mfma_32x32x2(float a, float b, f32x16_t c) {
  int slot = wave_deposit(a, b);
  State& s = S();
  int me = s.cur->lin, w = me / WAVE, lane = me % WAVE;
  WaveXchg& x = s.waves[w];
  int col = lane & 31, hi = lane >> 5;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    for (int k = 0; k < 2; ++k) acc = fmaf(x.a[slot][k * 32 + row], x.b[slot][k * 32 + col], acc);
    c[r] = acc;
  }
  return c;
}

inline f32x4_t mfma_16x16x4(float a, float b, f32x4_t c) {
  int slot = wave_deposit(a, b);
  State& s = S();
  int me = s.cur->lin, w = me / WAVE, lane = me % WAVE;
  WaveXchg& x = s.waves[w];
  int col = lane & 15, q = lane >> 4;
  for (int r = 0; r < 4; ++r) {
    int row = q * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(x.a[slot][k * 16 + row], x.b[slot][k * 16 + col], acc);
    c[r] = acc;
  }
  return c;
}

inline float shfl_idx(float v, int src_lane) {
  int slot = wave_deposit(v, 0.f);
  State& s = S();
  int w = s.cur->lin / WAVE;
  return s.waves[w].a[slot][src_lane & 63];
}

inline void fiber_entry() {
  State& s = S();
  s.body();
  s.cur->done = true;
  swapcontext(&s.cur->ctx, &s.sched);
}

inline void launch(dim3 grid, dim3 block, size_t shmem, std::function<void()> body) {
  State& s = S();
  int nt = block.x * block.y * block.z;
  s.nthreads = nt;
  s.blockDim = block;
  s.gridDim = grid;
  s.body = body;
  if ((int)s.fibers.size() < nt) {
    s.fibers.resize(nt);
    for (auto& f : s.fibers)
      if (!f.stack) f.stack = (char*)malloc(STACK);
  }
  std::vector<char> smem(shmem + 64, 0);
  s.dyn_smem = (void*)(((uintptr_t)smem.data() + 15) & ~(uintptr_t)15);
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        s.blockIdx = dim3(bx, by, bz);
        s.barrier_arrive = 0;
        s.waves.assign((nt + WAVE - 1) / WAVE, WaveXchg());
        s.wave_gen.assign(nt, 0);
        s.bar_gen.assign(nt, 0);
        // poison dynamic LDS with NaN so that reads of unstaged LDS are caught
        {
          uint32_t* p = (uint32_t*)s.dyn_smem;
          for (size_t i = 0; i < shmem / 4; ++i) p[i] = 0x7fc00000u;
        }
        for (int t = 0; t < nt; ++t) {
          Fiber& f = s.fibers[t];
          f.done = false;
          f.lin = t;
          f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = f.stack;
          f.ctx.uc_stack.ss_size = STACK;
          f.ctx.uc_link = &s.sched;
          makecontext(&f.ctx, (void (*)())fiber_entry, 0);
        }
        int remaining = nt;
        unsigned long long spins = 0;
        while (remaining > 0) {
          remaining = 0;
          for (int t = 0; t < nt; ++t) {
            Fiber& f = s.fibers[t];
            if (f.done) continue;
            s.cur = &f;
            swapcontext(&s.sched, &f.ctx);
            if (!f.done) remaining++;
          }
          if (++spins > 50000000ull) {
            fprintf(stderr, "hipsim: deadlock (divergent barrier / wave op?)\n");
            abort();
          }
        }
      }
  s.dyn_smem = nullptr;
}

}  // namespace sim

// ---- HIP spellings ----------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__
#define threadIdx (sim::S().cur->tid)
#define blockIdx (sim::S().blockIdx)
#define blockDim (sim::S().blockDim)
#define gridDim (sim::S().gridDim)
#define __syncthreads() sim::block_barrier()

static inline float __shfl_xor(float v, int m) {
  int lane = sim::S().cur->lin % 64;
  return sim::shfl_idx(v, lane ^ m);
}
static inline float __shfl_down(float v, int d) {
  int lane = sim::S().cur->lin % 64;
  int src = lane + d;
  return sim::shfl_idx(v, src < 64 ? src : lane);
}
static inline float __shfl(float v, int src) { return sim::shfl_idx(v, src); }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
static inline unsigned long long atomicExch(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = v; return o; }
static inline int atomicOr(int* p, int v) { int o = *p; *p = o | v; return o; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline void __threadfence() {}
static inline unsigned __brev(unsigned x) {
  unsigned r = 0;
  for (int i = 0; i < 32; ++i) r |= ((x >> i) & 1u) << (31 - i);
  return r;
}
static inline unsigned __umulhi(unsigned a, unsigned b) {
  return (unsigned)(((unsigned long long)a * b) >> 32);
}

// ---- minimal runtime stubs ------------------------------------------------------------
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "sim"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
  memset(p, v, n);
  return 0;
}
static inline hipError_t hipMemcpyAsyncD2D(void* d, const void* s, size_t n, hipStream_t) {
  memcpy(d, s, n);
  return 0;
}
