// Probe: which compute units does a hipExtStreamCreateWithCUMask mask select on this part?
//   hipcc --offload-arch=gfx950 -O3 -o cu_mask cu_mask.hip && ./cu_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ void where(unsigned* out) {
  // keep the workgroup alive for a while so that the launch spreads over every CU it may use
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < 200000) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg(0xF804);          // HW_ID
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg(0xF814) & 7;  // XCC_ID
  }
}
static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t s;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: stream creation failed\n", name); return; }
  const int grid = 2048;
  unsigned* out; hipMalloc(&out, grid * 8);
  hipLaunchKernelGGL(where, dim3(grid), dim3(256), 64 * 1024, s, out);
  hipStreamSynchronize(s);
  std::vector<unsigned> h(grid * 2);
  hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::set<unsigned>> per_xcc;
  for (int i = 0; i < grid; ++i) per_xcc[h[2 * i + 1]].insert((h[2 * i] >> 8) & 0xFF);     // cu[11:8] sh[12] se[15:13]
  printf("%s:", name);
  int total = 0;
  for (auto& kv : per_xcc) { printf("  xcc%u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
  printf("  -> %d CUs\n", total);
  hipFree(out); hipStreamDestroy(s);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, words = (cus + 31) / 32;
  printf("multiProcessorCount %d\n", cus);
  std::vector<uint32_t> all(words, 0xffffffffu), lowhalf(words, 0), xcd_lo(words, 0), cu_even(words, 0), first32(words, 0), xcd0(words, 0);
  for (int i = 0; i < cus; ++i) {
    if (i < cus / 2) lowhalf[i >> 5] |= 1u << (i & 31);
    if ((i & 7) < 4) xcd_lo[i >> 5] |= 1u << (i & 31);
    if (((i >> 3) & 1) == 0) cu_even[i >> 5] |= 1u << (i & 31);
    if (i < 32) first32[i >> 5] |= 1u << (i & 31);
    if ((i & 7) == 0) xcd0[i >> 5] |= 1u << (i & 31);
  }
  run("all bits", all); run("bits 0..N/2-1", lowhalf); run("bits with (i & 7) < 4", xcd_lo); run("bits with ((i >> 3) & 1) == 0", cu_even);
  run("bits 0..31", first32); run("bits with (i & 7) == 0", xcd0);
  return 0;
}
