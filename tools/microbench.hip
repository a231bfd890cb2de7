// tools/microbench.hip -- primitive latencies in the decoder's launch geometry
// (one 256-thread workgroup per CU, 256 workgroups).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define R 256
__global__ void __launch_bounds__(256) mb(unsigned long long* out, unsigned long long* tab, int nIter) {
  __shared__ unsigned int lds[4096];
  __shared__ unsigned long long l64[64];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 256) lds[i] = (i * 7 + 1) & 4095;
  if (tid < 64) l64[tid] = 0;
  __syncthreads();
  unsigned long long t0, t1, acc[12] = {0};
  // 0: empty s_memtime pair
  t0 = __builtin_readcyclecounter(); t1 = __builtin_readcyclecounter(); acc[0] = t1 - t0;
  // 1: barrier
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < R; ++i) __syncthreads();
  t1 = __builtin_readcyclecounter(); acc[1] = (t1 - t0) / R;
  // 2: dependent LDS read chain
  unsigned int p = tid & 63;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < R; ++i) p = lds[p];
  t1 = __builtin_readcyclecounter(); acc[2] = (t1 - t0) / R;
  // 3: LDS atomic add returning, dependent
  unsigned int q = p & 1023;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < R; ++i) q = atomicAdd(&lds[(q & 1023) + 2048], 1u) & 1023;
  t1 = __builtin_readcyclecounter(); acc[3] = (t1 - t0) / R;
  // 4: ballot + readlane dependent
  unsigned int v = q;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < R; ++i) { unsigned long long m = __ballot(v & 1); v = __builtin_amdgcn_readlane(v, i & 63) + (unsigned)__popcll(m); }
  t1 = __builtin_readcyclecounter(); acc[4] = (t1 - t0) / R;
  // 5: f64 dependent add chain
  double d = (double)v;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < R; ++i) d = d + 1.000001;
  t1 = __builtin_readcyclecounter(); acc[5] = (t1 - t0) / R;
  // 6: u64 compare/select chain
  unsigned long long k = (unsigned long long)d;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < R; ++i) k = (k > 12345ull + i) ? k - 3 : k + 7;
  t1 = __builtin_readcyclecounter(); acc[6] = (t1 - t0) / R;
  // 7: global sc1 load dependent chain on a private 1 MB table
  unsigned long long* mytab = tab + (size_t)blockIdx.x * 131072;
  unsigned long long g = (k + tid) & 131071;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 32; ++i) g = (__hip_atomic_load(&mytab[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + g * 2654435761ull + 12345) & 131071;
  t1 = __builtin_readcyclecounter(); acc[7] = (t1 - t0) / 32;
  // 8: global atomic CAS dependent chain
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 32; ++i) g = (atomicCAS(&mytab[g], 0ull, 0ull) + g * 2654435761ull + 777) & 131071;
  t1 = __builtin_readcyclecounter(); acc[8] = (t1 - t0) / 32;
  // 9: global plain store + load same address (write-through check)
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 32; ++i) { mytab[g] = 0; g = (mytab[(g + 1) & 131071] + g * 2654435761ull + 99) & 131071; }
  t1 = __builtin_readcyclecounter(); acc[9] = (t1 - t0) / 32;
  // 10: 64-bit LDS atomic max
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < R; ++i) k = atomicMax(&l64[tid & 63], k + i);
  t1 = __builtin_readcyclecounter(); acc[10] = (t1 - t0) / R;
  // 11: integer division chain
  int dv = (int)(g & 1023) + 1000000;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < R; ++i) dv = dv / (nIter + (i & 1)) + 1000000;
  t1 = __builtin_readcyclecounter(); acc[11] = (t1 - t0) / R;
  if (tid == 0) for (int i = 0; i < 12; ++i) out[blockIdx.x * 12 + i] = acc[i] + (i == 11 ? (dv & 1) : 0) + (i == 10 ? (k & 0) : 0) + (i==5 ? ((long long)d & 0):0);
}
int main() {
  unsigned long long *out, *tab;
  hipMalloc(&out, 256 * 12 * 8); hipMalloc(&tab, 256ull * 131072 * 8); hipMemset(tab, 0, 256ull * 131072 * 8);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(mb, dim3(256), dim3(256), 0, 0, out, tab, 29); hipDeviceSynchronize(); }
  std::vector<unsigned long long> h(256 * 12); hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  const char* names[12] = {"memtime pair", "barrier(256thr)", "lds read dep", "lds atomic add ret", "ballot+readlane", "f64 add dep", "u64 cmp/sel dep", "global sc1 load dep", "global CAS dep", "global st+ld", "lds atomic max64", "int div"};
  for (int i = 0; i < 12; ++i) { double s = 0; for (int b = 0; b < 256; ++b) s += h[b * 12 + i]; printf("%-22s %8.1f clocks\n", names[i], s / 256); }
  return 0;
}
