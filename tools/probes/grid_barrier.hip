// grid_barrier.hip -- what does one grid-wide phase boundary cost inside a persistent kernel on MI355X?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/grid_barrier.hip -o tools/probes/grid_barrier
// The small-batch reverse chain (3 x 333 ... 3 x 1000 patients) is a sequence of ~12 tiny layers per step whose launches
// each cost 4-5 us of boundary (profiles/r04_refw_kernel_stats.csv: k_add_int 4.0 us, k_gn_reduce 4.8 us for no work to
// speak of).  The alternative is ONE resident kernel with a barrier over all workgroups between layers.  This probe times
// that barrier alone and with the traffic pattern of a layer around it (every workgroup writes a slab other workgroups read).
//   mode 0: monotonic counter, relaxed agent atomics, no fences            (the floor)
//   mode 1: + agent-scope release before the arrive, acquire after the wait (what a data-carrying barrier needs)
//   mode 2: mode 1 + each workgroup writes `kb` KiB before the barrier and reads `kb` KiB another workgroup wrote after it
//   mode 3: the slab exchange with agent-scope (sc1) stores and loads instead of fences: nothing but s_waitcnt vmcnt(0) before the arrive
//   barrier 1 (second column block): two-level -- workgroups arrive on their XCD's counter (blockIdx % 8), the last of an XCD on the
//   global one, the last of all publishes the generation in 8 per-XCD words; everyone polls only its XCD's word
// Every spin is bounded (s_memrealtime budget): the kernel always drains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Args { int hier; unsigned* counter; unsigned* status; float* buf; int iters; int mode; int floats_per_wg; unsigned long long budget; unsigned long long* cyc; };

__device__ __forceinline__ bool grid_sync(unsigned* counter, unsigned* status, unsigned want, unsigned long long budget, bool fences) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x < 64) {                     // wave 0, wave-uniform control flow
    if (fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
      const unsigned v = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if (v >= want) break;
      if ((unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0u) { ok = false; break; }
      if (__builtin_amdgcn_s_memrealtime() - t0 > budget) { if (threadIdx.x == 0) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok;
}

// two-level barrier: counter[64 * x] per-XCD arrivals (x = 0..7), counter[64 * 8] global, counter[64 * (9 + x)] generation words
__device__ __forceinline__ bool grid_sync2(unsigned* c, unsigned* status, unsigned gen, unsigned long long budget, bool fences) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x < 64) {
    const unsigned x = blockIdx.x & 7u, per = (gridDim.x + 7u - x) / 8u;      // workgroups with this blockIdx % 8
    if (fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    unsigned last = 0;
    if (threadIdx.x == 0) {
      const unsigned v = __hip_atomic_fetch_add(c + 64 * x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v + 1 == per * gen) {
        const unsigned w = __hip_atomic_fetch_add(c + 64 * 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned nx = gridDim.x < 8u ? gridDim.x : 8u;
        if (w + 1 == nx * gen) last = 1;
      }
    }
    last = (unsigned)__builtin_amdgcn_readfirstlane((int)last);
    if (last) {
      if (threadIdx.x < 8) __hip_atomic_store(c + 64 * (9 + threadIdx.x), gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        const unsigned v = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(c + 64 * (9 + x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (v >= gen) break;
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0u) { ok = false; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > budget) { if (threadIdx.x == 0) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_sc1(float* p, float4 v) {
  const f32x4 w = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ float4 ld_sc1(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return make_float4(v.x, v.y, v.z, v.w);
}

__global__ __launch_bounds__(256) void k_barrier(Args a) {
  const unsigned G = gridDim.x;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  float acc = 0.f;
  unsigned bad = 0;
  for (int it = 0; it < a.iters; ++it) {
    if (a.mode == 3) {
      float* mine = a.buf + (size_t)((it & 1) * G + blockIdx.x) * a.floats_per_wg;
      for (int i = threadIdx.x * 4; i < a.floats_per_wg; i += 1024) st_sc1(mine + i, make_float4((float)it, (float)blockIdx.x, 1.f, 2.f));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (a.mode == 2) {
      float* mine = a.buf + (size_t)((it & 1) * G + blockIdx.x) * a.floats_per_wg;
      for (int i = threadIdx.x * 4; i < a.floats_per_wg; i += 1024)
        *reinterpret_cast<float4*>(mine + i) = make_float4((float)it, (float)blockIdx.x, 1.f, 2.f);
    }
    const bool fences = a.mode == 1 || a.mode == 2;
    if (a.hier ? !grid_sync2(a.counter, a.status, (unsigned)(it + 1), a.budget, fences)
               : !grid_sync(a.counter, a.status, G * (unsigned)(it + 1), a.budget, fences)) break;
    if (a.mode == 3) {
      const unsigned other = (blockIdx.x * 37u + 101u) % G;
      const float* theirs = a.buf + (size_t)((it & 1) * G + other) * a.floats_per_wg;
      for (int i = threadIdx.x * 4; i < a.floats_per_wg; i += 1024) {
        const float4 v = ld_sc1(theirs + i);
        acc += v.z;
        bad += (v.x != (float)it) | (v.y != (float)other);
      }
    }
    if (a.mode == 2) {
      const unsigned other = (blockIdx.x * 37u + 101u) % G;        // a workgroup of (most likely) another XCD
      const float* theirs = a.buf + (size_t)((it & 1) * G + other) * a.floats_per_wg;
      for (int i = threadIdx.x * 4; i < a.floats_per_wg; i += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(theirs + i);
        acc += v.z;
        bad += (v.x != (float)it) | (v.y != (float)other);
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { a.cyc[blockIdx.x * 2] = c1 - c0; }
  if (acc == -1.f) a.cyc[0] = 0;
  if (bad) atomicAdd(reinterpret_cast<unsigned*>(a.cyc + blockIdx.x * 2 + 1), bad);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned *counter, *status; float* buf; unsigned long long* cyc;
  const int maxG = 1024, maxf = 64 * 1024;
  CK(hipMalloc(&counter, 8192)); CK(hipMalloc(&status, 256));
  CK(hipMalloc(&buf, (size_t)2 * maxG * maxf * 4)); CK(hipMalloc(&cyc, maxG * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int Gs[] = {64, 128, 256, 512};
  const int kbs[] = {0, 0, 16, 64, 16, 64, 256};
  const int modes[] = {0, 1, 2, 2, 3, 3, 3};
  printf("%-5s %-6s %-5s %-7s %10s %10s %s\n", "hier", "G", "mode", "KiB/wg", "us/iter", "status", "stale reads");
  for (int hier = 0; hier < 2; ++hier)
  for (int G : Gs)
    for (int v = 0; v < 7; ++v) {
      Args a{hier, counter, status, buf, iters, modes[v], kbs[v] * 256, 100000000ull / 10 /* 100 ms at 100 MHz */, cyc};
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(counter, 0, 8192)); CK(hipMemset(status, 0, 256)); CK(hipMemset(cyc, 0, maxG * 16));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_barrier, dim3(G), dim3(256), 0, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 1) {
          unsigned st; CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
          std::vector<unsigned long long> h(G * 2); CK(hipMemcpy(h.data(), cyc, G * 16, hipMemcpyDeviceToHost));
          unsigned long long bad = 0; for (int i = 0; i < G; ++i) bad += h[2 * i + 1];
          printf("%-5d %-6d %-5d %-7d %10.3f %10u %llu\n", hier, G, modes[v], kbs[v], ms * 1000.0 / iters, st, bad);
        }
      }
    }
  return 0;
}
