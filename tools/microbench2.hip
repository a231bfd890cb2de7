// tools/microbench2.hip -- issue-rate / latency of the primitives the frame step is made of,
// at 1 and 2 waves per SIMD (256 / 512 threads per workgroup, one workgroup per CU).
// hipcc --offload-arch=gfx950 -O3 tools/microbench2.hip -o /tmp/mb2 && /tmp/mb2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define R 512
#define NT 16
static __device__ __forceinline__ unsigned long long clk_() {
  unsigned long long t;
  __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define CLK() clk_()
#define KEEP32(x) __asm__ volatile("" : "+v"(x))
#define KEEP64(x) __asm__ volatile("" : "+v"(x))
template <int W>
__global__ void __launch_bounds__(W) mb(unsigned long long* out, int one, double done) {
  __shared__ __attribute__((aligned(16))) unsigned int lds[8192];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += W) lds[i] = (i * 7 + 1) & 1023;
  __syncthreads();
  unsigned long long t0, t1, acc[NT] = {0};
  // 0: u32 add dependent chain
  unsigned int a = tid;
  t0 = CLK();
#pragma unroll 16
  for (int i = 0; i < R; ++i) a = a * 3u + (unsigned)one;
  KEEP32(a);
  t1 = CLK(); acc[0] = t1 - t0;
  // 1: f64 add dependent chain
  double d = (double)a;
  t0 = CLK();
#pragma unroll 16
  for (int i = 0; i < R; ++i) d = d + done;
  KEEP64(d);
  t1 = CLK(); acc[1] = t1 - t0;
  // 2: 4 independent f64 chains
  double d0 = d, d1 = d + 1, d2 = d + 2, d3 = d + 3;
  t0 = CLK();
#pragma unroll 4
  for (int i = 0; i < R; ++i) { d0 += done; d1 += done; d2 += done; d3 += done; }
  KEEP64(d0); KEEP64(d1); KEEP64(d2); KEEP64(d3);
  t1 = CLK(); acc[2] = (t1 - t0) / 4;
  d = d0 + d1 + d2 + d3;
  // 3: f64 -> f32 cvt + back, dependent
  t0 = CLK();
#pragma unroll 16
  for (int i = 0; i < R; ++i) { float f = (float)d; d = (double)f + done; }
  KEEP64(d);
  t1 = CLK(); acc[3] = (t1 - t0) / 3;
  // 4: f64 compare + select chain
  t0 = CLK();
#pragma unroll 16
  for (int i = 0; i < R; ++i) d = (d > done * i) ? d - done : d + done;
  KEEP64(d);
  t1 = CLK(); acc[4] = (t1 - t0) / 4;
  // 5: DPP inclusive scan (6 steps + nothing else), dependent
  unsigned int s = a;
  t0 = CLK();
  for (int i = 0; i < R / 4; ++i) {
    unsigned int x = s;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, true);
    s = x;
  }
  KEEP32(s);
  t1 = CLK(); acc[5] = (t1 - t0) / (R / 4);
  // 6: ballot -> popcount -> broadcast dependent
  unsigned int v = s;
  t0 = CLK();
  for (int i = 0; i < R / 4; ++i) { unsigned long long m = __ballot(v & 1); v = v + (unsigned)__popcll(m); }
  KEEP32(v);
  t1 = CLK(); acc[6] = (t1 - t0) / (R / 4);
  // 7: readlane dependent
  t0 = CLK();
  for (int i = 0; i < R / 4; ++i) v = __builtin_amdgcn_readlane(v, 5) + v;
  KEEP32(v);
  t1 = CLK(); acc[7] = (t1 - t0) / (R / 4);
  // 8: LDS read b32 dependent
  unsigned int p = (tid * 17) & 1023;
  t0 = CLK();
  for (int i = 0; i < R / 4; ++i) p = lds[p];
  KEEP32(p);
  t1 = CLK(); acc[8] = (t1 - t0) / (R / 4);
  // 9: LDS read b128 x2 then use, dependent on the previous
  t0 = CLK();
  for (int i = 0; i < R / 4; ++i) {
    const uint4 q0 = ((const uint4*)lds)[(p & 255) * 2], q1 = ((const uint4*)lds)[(p & 255) * 2 + 1];
    p = (q0.x + q0.y + q0.z + q0.w + q1.x + q1.y + q1.z + q1.w) & 1023;
  }
  KEEP32(p);
  t1 = CLK(); acc[9] = (t1 - t0) / (R / 4);
  // 10: barrier (LDS-only wait)
  t0 = CLK();
  for (int i = 0; i < R / 4; ++i) __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  t1 = CLK(); acc[10] = (t1 - t0) / (R / 4);
  // 11: ds_add no return, distinct addresses per lane, then barrier
  t0 = CLK();
  for (int i = 0; i < R / 4; ++i) { atomicAdd(&lds[4096 + ((tid * 5 + i) & 511)], 1u); }
  __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  t1 = CLK(); acc[11] = (t1 - t0) / (R / 4);
  // 12: ds_add no return, all lanes of the workgroup same address
  t0 = CLK();
  for (int i = 0; i < R / 4; ++i) { atomicAdd(&lds[4096 + 600 + (i & 3)], 1u); }
  __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  t1 = CLK(); acc[12] = (t1 - t0) / (R / 4);
  // 13: 64-bit shift by lane-varying amount + and + cmp chain
  unsigned long long k = ((unsigned long long)v << 32) | p;
  t0 = CLK();
#pragma unroll 16
  for (int i = 0; i < R; ++i) k = ((k >> (lane & 63)) & 1ull) ? k * 3ull + 1ull : k + 7ull;
  KEEP64(k);
  t1 = CLK(); acc[13] = (t1 - t0) / 4;
  // 14: u32 v_med3 / shifts (the bin computation), dependent
  unsigned int bq = (unsigned)k;
  t0 = CLK();
#pragma unroll 16
  for (int i = 0; i < R; ++i) { int q = (int)(bq >> 15) - one; q = q < 0 ? 0 : q; bq = bq * 9u + (q > 511 ? 511 : q); }
  KEEP32(bq);
  t1 = CLK(); acc[14] = (t1 - t0) / 4;
  // 15: memtime pair
  t0 = CLK(); t1 = CLK(); acc[15] = t1 - t0;
  if (lane == 0) {
    for (int i = 0; i < NT; ++i) out[((size_t)blockIdx.x * (W / 64) + (tid >> 6)) * NT + i] = acc[i] + (i == 1 ? ((long long)d & 0) : 0) + (i == 13 ? (k & 0) : 0) + (i == 14 ? (bq & 0) : 0) + (i == 7 ? (v & 0) : 0);
  }
}
template <int W>
void run(const char* title) {
  unsigned long long* out;
  const int nb = 256, nw = W / 64;
  hipMalloc(&out, (size_t)nb * nw * NT * 8);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(mb<W>, dim3(nb), dim3(W), 0, 0, out, 1, 1.0000001); hipDeviceSynchronize(); }
  std::vector<unsigned long long> h((size_t)nb * nw * NT);
  hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  const char* names[NT] = {"u32 mad dep (per op)", "f64 add dep (per op)", "f64 add 4 chains (per op)", "f64<->f32 cvt+add dep (per op, 3 ops)", "f64 cmp+sel+add (per op, 4 ops)",
    "DPP scan 6 steps (per scan)", "ballot+popc+add dep (per iter)", "readlane+add dep (per iter)", "LDS read b32 dep", "LDS read 2xb128 + 8 adds dep", "lgkm wait + s_barrier",
    "ds_add distinct (per instr)", "ds_add same address (per instr)", "u64 shift/and/sel (per op, 4 ops)", "bin calc (per op, 4 ops)", "memtime pair"};
  const double div[NT] = {R, R, R, R, R, 1, 1, 1, 1, 1, 1, 1, 1, R, R, 1};
  printf("---- %s\n", title);
  for (int i = 0; i < NT; ++i) { double s = 0; for (size_t b = 0; b < (size_t)nb * nw; ++b) s += h[b * NT + i]; printf("%-40s %8.1f clocks\n", names[i], s / (nb * nw) / div[i]); }
  hipFree(out);
}
int main() { run<64>("64 threads (1 wave per CU)"); run<256>("256 threads (1 wave per SIMD)"); run<512>("512 threads (2 waves per SIMD)"); return 0; }
