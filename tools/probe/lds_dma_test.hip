#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* g, float* out, int n) {
  __shared__ float buf[3][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave == 1) {
    for (int r = 0; r < 3; ++r) {
      if (lane < n) {
        __builtin_amdgcn_global_load_lds(g + r * n + lane, &buf[r][0], 4, 0, 0);
      }
    }
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (threadIdx.x < 64) for (int r = 0; r < 3; ++r) out[r * 64 + threadIdx.x] = buf[r][threadIdx.x];
}
int main() {
  const int n = 29;
  float h[3 * 29], *g, *o, ho[192];
  for (int i = 0; i < 87; ++i) h[i] = 1000.f + i;
  hipMalloc(&g, sizeof(h)); hipMalloc(&o, sizeof(ho)); hipMemset(o, 0, sizeof(ho));
  hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, g, o, n);
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int r = 0; r < 3; ++r) for (int i = 0; i < n; ++i) if (ho[r * 64 + i] != h[r * n + i]) ++bad;
  printf("bad %d  sample %g %g %g\n", bad, ho[0], ho[64 + 5], ho[128 + 28]);
  return bad;
}
