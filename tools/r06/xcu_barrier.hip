// tools/r06/xcu_barrier.hip -- what a barrier between workgroups on DIFFERENT CUs costs on MI355X, and what it costs to
// read what another CU wrote before it (the two prices a multi-CU engine for one utterance would pay per phase).
//   hipcc --offload-arch=gfx950 -O3 tools/r06/xcu_barrier.hip -o /tmp/xcu_barrier && /tmp/xcu_barrier
// G workgroups of 1024 threads take part; `same_xcd` = 1 launches 8 x G workgroups and lets only those whose id is a
// multiple of 8 work (consecutive workgroup ids go round the 8 XCDs: the survivors share one XCD, i.e. one L2).
// The barrier: thread 0 of every workgroup adds to a counter in HBM (agent scope, release), spins until the counter
// says everybody has (acquire), then the workgroup's own barrier.  Payload variant: before the barrier every workgroup
// writes `bytes` to its slice, after it reads the NEXT workgroup's slice (coherent loads: past the L1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

static __device__ __forceinline__ unsigned long long clk_() {
  unsigned long long t;
  __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

__global__ void __launch_bounds__(1024) k(unsigned int* counter, unsigned int* data, unsigned long long* out, int G, int stride,
                                          int iters, int words) {
  const int wg = (int)blockIdx.x / stride;
  if ((int)blockIdx.x % stride != 0 || wg >= G) {
    return;
  }
  const int tid = (int)threadIdx.x;
  unsigned int acc = 0;
  __syncthreads();
  const unsigned long long t0 = clk_();
  for (int it = 0; it < iters; ++it) {
    for (int i = tid; i < words; i += 1024) { /* what this phase produced */
      data[(size_t)wg * words + i] = (unsigned)(it * 131 + i);
    }
    __threadfence(); /* agent-scope release: the stores are in L2 */
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int want = (unsigned)G * (unsigned)(it + 1);
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    const int nb = (wg + 1) % G;
    for (int i = tid; i < words; i += 1024) { /* ... is what the next phase of another CU reads */
      acc += __hip_atomic_load(&data[(size_t)nb * words + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  const unsigned long long t1 = clk_();
  if (tid == 0) {
    out[wg] = (t1 - t0) / (unsigned long long)iters;
    out[64 + wg] = acc;
  }
}

int main() {
  unsigned int *counter, *data;
  unsigned long long* out;
  hipMalloc(&counter, 4);
  hipMalloc(&data, 64u << 20);
  hipMalloc(&out, 8 * 128);
  std::vector<unsigned long long> h(128);
  const int iters = 2000;
  printf("# shader clocks (s_memtime: the counter the kernels' phase profiles use, ~2.4 GHz) per barrier round, slowest workgroup\n");
  printf("%-9s %-4s %-10s %-12s %-12s\n", "placement", "G", "payload_B", "clocks/round", "~us/round");
  for (int same = 1; same >= 0; --same) {
    for (int G : {2, 4, 8, 16, 32}) {
      for (int bytes : {0, 4096, 65536}) {
        hipMemset(counter, 0, 4);
        const int stride = same ? 8 : 1;
        k<<<dim3(G * stride), dim3(1024)>>>(counter, data, out, G, stride, iters, bytes / 4);
        if (hipDeviceSynchronize() != hipSuccess) {
          printf("launch failed\n");
          return 1;
        }
        hipMemcpy(h.data(), out, 8 * 128, hipMemcpyDeviceToHost);
        unsigned long long mx = 0;
        for (int i = 0; i < G; ++i) mx = h[i] > mx ? h[i] : mx;
        printf("%-9s %-4d %-10d %-12llu %-12.2f\n", same ? "one-XCD" : "any-XCD", G, bytes, mx, (double)mx / 2400.0);
      }
    }
  }
  return 0;
}
