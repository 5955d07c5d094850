// hipemu -- a tiny HOST emulation of the HIP device model.   TEST INFRASTRUCTURE ONLY.
//
// Purpose: this container has no GPU and every GPU call costs budgeted minutes, so the CPU test
// suite compiles the *unmodified* kernel sources of lemo_amd/csrc with a host compiler against
// this header (it shadows <hip/hip_runtime.h> on the include path) and checks index arithmetic,
// MFMA operand / accumulator lane maps and host orchestration against the oracle before any GPU
// time is spent.  It is NOT a product path: lemo_amd never loads the emulated library, the
// product loader only opens liblemo_hip.so built by hipcc for gfx950.
//
// Model: one block at a time; every thread of the block is a ucontext fiber on ONE OS thread,
// scheduled round-robin; __syncthreads / cross-lane ops / MFMA are generation barriers that yield.
// Wavefront = 64 lanes.  MFMA f32 lane maps follow /opt/skills/guides/cdna_hip_programming.md §3:
//   32x32x2 : A[i=l&31][k=l>>5]  B[k=l>>5][j=l&31]  D: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5)
//   16x16x4 : A[i=l&15][k=l>>4]  B[k=l>>4][j=l&15]  D: col=l&15, row=4*(l>>4)+r
// and numerics are a k-ordered fmaf chain (bit-exact per the guide).
#pragma once
#include <sys/mman.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __shared__ static
#define LEMO_PIN(x) ((void)0)
#define __launch_bounds__(...)
#define __constant__ static

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }

typedef void* hipStream_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotSupported = 801 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256); return *p ? hipSuccess : 2; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*) { return hipErrorNotSupported; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return hipErrorNotSupported; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
static inline hipError_t hipGraphUpload(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }

namespace hipemu {

constexpr int WAVE = 64;
constexpr size_t STACK = 256 * 1024;

struct Group { int count = 0, arrived = 0; unsigned gen = 0; };
struct WaveBuf {
  Group g;
  float a[WAVE], b[WAVE];
  float c[WAVE][16];
  unsigned long long u[WAVE];
  float a8[WAVE][8], b8[WAVE][8];
};
// Minimal x86-64 fiber switch (callee-saved registers + stack pointer).  glibc's swapcontext makes a
// sigprocmask syscall per switch, which dominated the emulation time (2 switches per lane per MFMA).
struct Ctx { void* sp = nullptr; };
extern "C" void hipemu_switch(Ctx* from, Ctx* to);
asm(R"(
.text
.weak hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq (%rsi), %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch, .-hipemu_switch
)");
struct Fiber { Ctx ctx; bool done = false; dim3 tid; int wave = 0, lane = 0; };

struct State {
  std::vector<Fiber> fibers;
  std::vector<WaveBuf> waves;
  std::vector<char*> stacks;
  Group block;
  Ctx sched;
  int cur = -1;
  std::function<void()> body;
};
inline State& st() { static State s; return s; }

}  // namespace hipemu

// threadIdx / blockIdx are plain globals rewritten by the scheduler before every resume
inline dim3& hipemu_threadIdx() { static dim3 v; return v; }
inline dim3& hipemu_blockIdx() { static dim3 v; return v; }
inline dim3& hipemu_blockDim() { static dim3 v; return v; }
inline dim3& hipemu_gridDim() { static dim3 v; return v; }
#define threadIdx (hipemu_threadIdx())
#define blockIdx (hipemu_blockIdx())
#define blockDim (hipemu_blockDim())
#define gridDim (hipemu_gridDim())

namespace hipemu {

inline void yield() {
  State& s = st();
  Fiber& f = s.fibers[s.cur];
  hipemu_switch(&f.ctx, &s.sched);
}
inline void barrier(Group& g) {
  unsigned gen = g.gen;
  if (++g.arrived == g.count) { g.arrived = 0; g.gen++; }
  else while (g.gen == gen) yield();
}
inline Fiber& me() { State& s = st(); return s.fibers[s.cur]; }
inline WaveBuf& mywave() { State& s = st(); return s.waves[s.fibers[s.cur].wave]; }

inline void fiber_entry() {
  State& s = st();
  s.body();
  s.fibers[s.cur].done = true;
  hipemu_switch(&s.fibers[s.cur].ctx, &s.sched);
  abort();                                                   // a finished fiber is never resumed
}

inline std::vector<char>& dyn_smem_buf() { static std::vector<char> b; return b; }
inline void* dyn_smem() { return (void*)(((uintptr_t)dyn_smem_buf().data() + 63) & ~(uintptr_t)63); }

inline void launch(dim3 grid, dim3 block, size_t shmem, std::function<void()> body) {
  State& s = st();
  if (dyn_smem_buf().size() < shmem + 64) dyn_smem_buf().resize(shmem + 64);
  const int nthr = (int)(block.x * block.y * block.z);
  const int nw = (nthr + WAVE - 1) / WAVE;
  while ((int)s.stacks.size() < nthr) {
    void* p = mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { perror("hipemu mmap"); abort(); }
    s.stacks.push_back((char*)p);
  }
  s.body = std::move(body);
  hipemu_blockDim() = block;
  hipemu_gridDim() = grid;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        s.fibers.assign(nthr, Fiber());
        s.waves.assign(nw, WaveBuf());
        s.block = Group();
        s.block.count = nthr;
        for (int t = 0; t < nthr; ++t) {
          Fiber& f = s.fibers[t];
          f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
          f.wave = t / WAVE;
          f.lane = t % WAVE;
          s.waves[f.wave].g.count++;
          // initial frame: 6 callee-saved slots + return address = fiber_entry; after the `ret` the stack
          // pointer is == 8 (mod 16), as at any function entry
          uintptr_t top = ((uintptr_t)s.stacks[t] + STACK) & ~(uintptr_t)15;
          void** sp = (void**)(top - 8) - 7;
          for (int q = 0; q < 6; ++q) sp[q] = nullptr;
          sp[6] = (void*)&fiber_entry;
          f.ctx.sp = sp;
        }
        int alive = nthr;
        while (alive > 0) {
          for (int t = 0; t < nthr; ++t) {
            Fiber& f = s.fibers[t];
            if (f.done) continue;
            s.cur = t;
            hipemu_threadIdx() = f.tid;
            hipemu_blockIdx() = dim3(bx, by, bz);
            hipemu_switch(&s.sched, &f.ctx);
            if (f.done) --alive;
          }
        }
      }
}

}  // namespace hipemu

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipemu::launch((grid), (block), (size_t)(shmem), [=]() { kernel(__VA_ARGS__); })
#define LEMO_DYN_SMEM(var) float* var = (float*)hipemu::dyn_smem()

// ---- synchronisation & cross-lane ---------------------------------------------------------
static inline void __syncthreads() { hipemu::barrier(hipemu::st().block); }
static inline void __builtin_amdgcn_s_barrier() { hipemu::barrier(hipemu::st().block); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

template <typename T>
static inline T hipemu_exchange(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "exchange <= 8 bytes");
  hipemu::WaveBuf& w = hipemu::mywave();
  const int lane = hipemu::me().lane;
  unsigned long long raw = 0;
  memcpy(&raw, &v, sizeof(T));
  w.u[lane] = raw;
  hipemu::barrier(w.g);
  unsigned long long got = (src_lane >= 0 && src_lane < w.g.count) ? w.u[src_lane] : raw;
  hipemu::barrier(w.g);
  T out;
  memcpy(&out, &got, sizeof(T));
  return out;
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) {
  const int lane = hipemu::me().lane;
  const int src = lane ^ mask;
  return hipemu_exchange(v, (src / width == lane / width) ? src : lane);
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
  const int lane = hipemu::me().lane;
  const int src = lane + (int)d;
  return hipemu_exchange(v, (src / width == lane / width) ? src : lane);
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
  const int lane = hipemu::me().lane;
  const int src = lane - (int)d;
  return hipemu_exchange(v, (src >= 0 && src / width == lane / width) ? src : lane);
}
template <typename T> static inline T __shfl(T v, int src, int width = 64) {
  const int lane = hipemu::me().lane;
  return hipemu_exchange(v, (lane / width) * width + (src % width));
}
static inline unsigned long long __ballot(int pred) {
  hipemu::WaveBuf& w = hipemu::mywave();
  const int lane = hipemu::me().lane;
  w.u[lane] = pred ? 1ull : 0ull;
  hipemu::barrier(w.g);
  unsigned long long m = 0;
  for (int i = 0; i < w.g.count; ++i) m |= (w.u[i] & 1ull) << i;
  hipemu::barrier(w.g);
  return m;
}

// DPP row operations used by the reductions (common.hpp): quad_perm, row_mirror, row_half_mirror, row_ror
static inline int __builtin_amdgcn_update_dpp(int, int src, int ctrl, int, int, bool) {
  const int lane = hipemu::me().lane;
  int from = lane;
  if (ctrl >= 0 && ctrl <= 0xFF) from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
  else if (ctrl == 0x140) from = (lane & ~15) | (15 - (lane & 15));
  else if (ctrl == 0x141) from = (lane & ~7) | (7 - (lane & 7));
  else if (ctrl > 0x120 && ctrl <= 0x12F) from = (lane & ~15) | ((lane - (ctrl - 0x120)) & 15);
  else abort();
  return hipemu_exchange(src, from);
}
static inline int __builtin_amdgcn_readlane(int v, int lane) { return hipemu_exchange(v, lane); }

// ---- MFMA (f32 in / f32 acc) ----------------------------------------------------------------
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));

static inline hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, hipemu_f32x16 c, int, int, int) {
  hipemu::WaveBuf& w = hipemu::mywave();
  const int l = hipemu::me().lane;
  w.a[l] = a; w.b[l] = b;
  hipemu::barrier(w.g);
  hipemu_f32x16 d;
  const int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    for (int k = 0; k < 2; ++k) acc = fmaf(w.a[row + 32 * k], w.b[col + 32 * k], acc);
    d[r] = acc;
  }
  hipemu::barrier(w.g);
  return d;
}
static inline hipemu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
  hipemu::WaveBuf& w = hipemu::mywave();
  const int l = hipemu::me().lane;
  w.a[l] = a; w.b[l] = b;
  hipemu::barrier(w.g);
  hipemu_f32x4 d;
  const int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * (l >> 4) + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(w.a[row + 16 * k], w.b[col + 16 * k], acc);
    d[r] = acc;
  }
  hipemu::barrier(w.g);
  return d;
}

// gfx950 bf16 MFMA: A[i = l&31][k = 8*(l>>5)+e], B[k][j = l&31], C/D as the f32 32x32 form; the 16
// products of one instruction are exact and summed before the single fp32 rounding
typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));
static inline hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x16 c, int, int, int) {
  hipemu::WaveBuf& w = hipemu::mywave();
  const int l = hipemu::me().lane;
  for (int e = 0; e < 8; ++e) { w.a8[l][e] = (float)a[e]; w.b8[l][e] = (float)b[e]; }
  hipemu::barrier(w.g);
  hipemu_f32x16 d;
  const int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    double acc = c[r];
    for (int hh = 0; hh < 2; ++hh)
      for (int e = 0; e < 8; ++e) acc += (double)w.a8[row + 32 * hh][e] * (double)w.b8[col + 32 * hh][e];
    d[r] = (float)acc;
  }
  hipemu::barrier(w.g);
  return d;
}

// ---- atomics (single OS thread => plain RMW is atomic w.r.t. other fibers) --------------------
template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline float atomicAdd(float* p, double v) { float o = *p; *p = o + (float)v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; *p = std::max(o, v); return o; }
template <typename T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }

// ---- buffer resources / cache-policy loads & stores / scoped atomics (plain memory on the host) ------------
struct hipDeviceProp_t { int multiProcessorCount; };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
// (workgroups run one after another here: multi-layer chains cannot make progress -- the CPU tests use n = 1)
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 1 << 20; return hipSuccess; }
struct hipemu_rsrc { char* p; };
typedef hipemu_rsrc __amdgpu_buffer_rsrc_t;
typedef unsigned hipemu_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned hipemu_u32x2 __attribute__((ext_vector_type(2)));
static inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short, int, int) { return hipemu_rsrc{(char*)p}; }
static inline hipemu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
  hipemu_u32x4 v; memcpy(&v, r.p + voff + soff, 16); return v;
}
static inline unsigned __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
  unsigned v; memcpy(&v, r.p + voff + soff, 4); return v;
}
static inline void __builtin_amdgcn_raw_buffer_store_b128(hipemu_u32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int) { memcpy(r.p + voff + soff, &v, 16); }
static inline void __builtin_amdgcn_raw_buffer_store_b64(hipemu_u32x2 v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int) { memcpy(r.p + voff + soff, &v, 8); }
static inline void __builtin_amdgcn_raw_buffer_store_b32(unsigned v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int) { memcpy(r.p + voff + soff, &v, 4); }
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
template <typename T> static inline T hipemu_fetch_add(T* p, T v) { T o = *p; *p = o + v; return o; }
#define __hip_atomic_fetch_add(p, v, order, scope) hipemu_fetch_add((p), (v))
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline void __builtin_amdgcn_s_waitcnt(int) {}

static inline void __builtin_amdgcn_wave_barrier() { hipemu::barrier(hipemu::mywave().g); }
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
// only ever applied to wave-uniform values in the kernels
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
static inline unsigned __builtin_amdgcn_s_getreg(int) { return 0u; }
static inline unsigned long long __builtin_amdgcn_s_memtime() { return 0ull; }

// ---- math -----------------------------------------------------------------------------------
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __fdividef(float a, float b) { return a / b; }
#define __expf(x) expf(x)
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
using std::max;
using std::min;
