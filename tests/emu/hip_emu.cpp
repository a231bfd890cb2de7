/* tests/emu/hip_emu.cpp -- see hip_emu.h.  TEST INFRASTRUCTURE ONLY. */
#include "hip_emu.h"

#include <cstdlib>

thread_local EmuDim threadIdx, blockIdx, blockDim;
thread_local EmuBlock* emuBlock;

void emuLaunch(int nBlocks, int W, size_t ldsBytes, const std::function<void(char*)>& fn) {
  for (int b = 0; b < nBlocks; ++b) {
    EmuBlock blk;
    pthread_barrier_init(&blk.bar, nullptr, (unsigned)W);
    blk.waves = std::vector<EmuWave>((size_t)(W + 63) / 64);
    for (auto& w : blk.waves) {
      pthread_barrier_init(&w.bar, nullptr, 64);
    }
    void* lds = nullptr;
    if (posix_memalign(&lds, 64, ldsBytes ? ldsBytes : 64) != 0) {
      abort();
    }
    memset(lds, 0xA5, ldsBytes ? ldsBytes : 64); /* LDS is uninitialised on a GPU */
    std::vector<std::thread> th;
    th.reserve(W);
    for (int t = 0; t < W; ++t) {
      th.emplace_back([&, t]() {
        threadIdx.x = (unsigned)t;
        blockIdx.x = (unsigned)b;
        blockDim.x = (unsigned)W;
        emuBlock = &blk;
        fn((char*)lds);
      });
    }
    for (auto& x : th) {
      x.join();
    }
    free(lds);
    pthread_barrier_destroy(&blk.bar);
    for (auto& w : blk.waves) {
      pthread_barrier_destroy(&w.bar);
    }
  }
}
