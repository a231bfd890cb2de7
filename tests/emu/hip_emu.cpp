/* tests/emu/hip_emu.cpp -- see hip_emu.h.  TEST INFRASTRUCTURE ONLY.
 *
 * One host thread per wave, the wave's 64 lanes as fibers on it.  A fiber switch saves the callee-saved registers
 * on the lane's own stack and swaps the stack pointer (x86-64 System V; no signal mask, no syscall).  Workgroups of
 * one launch run a few at a time (they are independent utterances, as on the device). */
#include "hip_emu.h"

#include <sys/mman.h>

#include <cstdio>
#include <cstdlib>

#if !defined(__x86_64__)
#error "the emulator's fiber switch is written for x86-64"
#endif

thread_local EmuDim threadIdx, blockIdx, blockDim;
thread_local EmuBlock* emuBlock;

extern "C" void emu_switch(void** saveSp, void* toSp);
__asm__(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_switch,.-emu_switch
)");

namespace {
constexpr size_t kStack = 1u << 20; /* per lane (reserved, committed on touch) */

thread_local EmuWave* curWave; /* the wave this host thread runs */

/* next lane after `me` that has not finished, or -1 */
int nextLane(const EmuWave& w, int me) {
  for (int k = 1; k <= 64; ++k) {
    const int nx = (me + k) & 63;
    if (!w.done[nx]) {
      return nx;
    }
  }
  return -1;
}

/* every lane starts here, on its own stack */
__attribute__((noinline, used)) void laneEntry() {
  EmuWave& w = *curWave;
  const int me = w.cur;
  threadIdx.x = w.base + (unsigned)me;
  (*emuBlock->fn)(emuBlock->lds);
  w.done[me] = true;
  ++w.nDone;
  const int nx = nextLane(w, me);
  void* dummy;
  if (nx < 0) {
    emu_switch(&dummy, w.schedSp); /* the wave has finished: back to its thread */
  } else {
    w.cur = nx;
    emu_switch(&dummy, w.sp[nx]);
  }
  abort(); /* (a finished lane is never resumed) */
}

void runWave(EmuBlock* blk, int wave, unsigned block, unsigned W, char* stacks) {
  EmuWave& w = blk->waves[(size_t)wave];
  emuBlock = blk;
  curWave = &w;
  blockIdx.x = block;
  blockDim.x = W;
  w.base = (unsigned)wave * 64u;
  w.cur = 0;
  w.nDone = 0;
  w.waveGen = w.blockGen = 0u;
  w.waveArrive = w.blockArrive = 0;
  for (int l = 0; l < 64; ++l) {
    w.done[l] = false;
    char* top = stacks + ((size_t)wave * 64 + (size_t)l + 1) * kStack;
    void** sp = (void**)((uintptr_t)top & ~(uintptr_t)15);
    *--sp = nullptr;                 /* where laneEntry would return to (it does not) */
    *--sp = (void*)&laneEntry;       /* emu_switch's ret */
    for (int k = 0; k < 6; ++k) {
      *--sp = nullptr;               /* rbp rbx r12 r13 r14 r15 */
    }
    w.sp[l] = (void*)sp;
  }
  emu_switch(&w.schedSp, w.sp[0]);
  if (w.nDone != 64) {
    fprintf(stderr, "hip_emu: wave %d returned with %d of 64 lanes finished\n", wave, w.nDone);
    abort();
  }
}
} // namespace

void emuYield() {
  EmuWave& w = *curWave;
  const int me = w.cur;
  const int nx = nextLane(w, me);
  if (nx < 0 || nx == me) {
    return;
  }
  const unsigned t = threadIdx.x;
  w.cur = nx;
  emu_switch(&w.sp[me], w.sp[nx]);
  threadIdx.x = t; /* (the lanes of a wave share the thread's threadIdx) */
}

void emuLaunch(int nBlocks, int W, size_t ldsBytes, const std::function<void(char*)>& fn) {
  if (W <= 0 || (W & 63) != 0) {
    fprintf(stderr, "hip_emu: %d threads per workgroup (whole waves only)\n", W);
    abort();
  }
  const int nWaves = W / 64;
  const char* env = getenv("FLTX_EMU_BLOCKS");
  int par = env ? atoi(env) : 4;
  par = par < 1 ? 1 : (par > nBlocks ? nBlocks : par);
  const size_t ldsAlloc = ((ldsBytes ? ldsBytes : 64) + 63) & ~(size_t)63;
  const size_t stackBytes = (size_t)W * kStack;
  struct Slot {
    EmuBlock blk;
    char* stacks;
    void* lds;
  };
  std::vector<Slot> slots((size_t)par);
  for (auto& s : slots) {
    s.stacks = (char*)mmap(nullptr, stackBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (s.stacks == (char*)MAP_FAILED || posix_memalign(&s.lds, 64, ldsAlloc) != 0) {
      fprintf(stderr, "hip_emu: no memory for the lanes' stacks\n");
      abort();
    }
    pthread_barrier_init(&s.blk.bar, nullptr, (unsigned)nWaves);
    s.blk.waves = std::vector<EmuWave>((size_t)nWaves);
    s.blk.fn = &fn;
    s.blk.lds = (char*)s.lds;
  }
  for (int b0 = 0; b0 < nBlocks; b0 += par) {
    const int n = nBlocks - b0 < par ? nBlocks - b0 : par;
    std::vector<std::thread> th;
    th.reserve((size_t)n * (size_t)nWaves);
    for (int i = 0; i < n; ++i) {
      Slot& s = slots[(size_t)i];
      memset(s.lds, 0xA5, ldsAlloc); /* LDS is uninitialised on a GPU */
      for (int wv = 0; wv < nWaves; ++wv) {
        th.emplace_back(runWave, &s.blk, wv, (unsigned)(b0 + i), (unsigned)W, s.stacks);
      }
    }
    for (auto& x : th) {
      x.join();
    }
  }
  for (auto& s : slots) {
    pthread_barrier_destroy(&s.blk.bar);
    munmap(s.stacks, stackBytes);
    free(s.lds);
  }
}
