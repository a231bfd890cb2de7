// tools/probe/memlat.hip -- latency of a dependent random 16-byte load, one wave per CU on every CU,
// by table size (what an n-gram probe of fltx_ylane.h costs).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
__global__ void chase(const uint4* tab, uint32_t mask, int steps, unsigned long long* out, int coherent) {
  uint32_t idx = (blockIdx.x * 2654435761u + threadIdx.x * 40503u) & mask;
  unsigned long long t0 = clock64();
  for (int i = 0; i < steps; ++i) {
    uint4 v;
    if (coherent) {
      v.x = __hip_atomic_load(&tab[idx].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      v = tab[idx];
    }
    idx = (v.x + i * 7919u) & mask;
  }
  unsigned long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = (t1 - t0) / steps + (idx & 0);
}
int main() {
  const size_t maxN = 1u << 24; // 16 M entries x 16 B = 256 MB
  uint4* tab; hipMalloc(&tab, maxN * 16);
  std::vector<uint4> h(maxN);
  uint32_t s = 12345;
  for (size_t i = 0; i < maxN; ++i) { s = s * 1664525u + 1013904223u; h[i] = {s >> 4, 0, 0, 0}; }
  hipMemcpy(tab, h.data(), maxN * 16, hipMemcpyHostToDevice);
  unsigned long long* out; hipMalloc(&out, 256 * 8);
  for (int coh = 0; coh < 2; ++coh)
  for (int lg = 12; lg <= 24; lg += 2) {
    const uint32_t mask = (1u << lg) - 1;
    for (int threads : {1, 64}) {
      hipLaunchKernelGGL(chase, dim3(256), dim3(threads), 0, 0, tab, mask, 200, out, coh);
      hipLaunchKernelGGL(chase, dim3(256), dim3(threads), 0, 0, tab, mask, 2000, out, coh);
      hipDeviceSynchronize();
      unsigned long long r[256]; hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
      double a = 0; for (int i = 0; i < 256; ++i) a += r[i];
      printf("%s table %8.2f MB, %2d lanes/wave active: %7.0f clocks per dependent load\n", coh ? "agent-coherent" : "plain", (double)(1u << lg) * 16 / 1e6, threads, a / 256);
    }
  }
  return 0;
}
