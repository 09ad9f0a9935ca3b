// pp_emu.cpp -- TEST INFRASTRUCTURE ONLY: fiber-based SPMD executor (see pp_emu.h).
#include "pp_emu.h"

#include <sys/mman.h>
#include <ucontext.h>

#include <atomic>
#include <thread>
#include <vector>

namespace pp_emu {

thread_local ThreadCtx* cur = nullptr;
thread_local unsigned char* dyn_smem = nullptr;

namespace {
constexpr size_t kStackBytes = 256 * 1024;
constexpr int kMaxThreads = 1024;
constexpr int kWave = 64;

struct Fiber {
  ucontext_t ctx;
  ThreadCtx tctx;
  bool done = false;
};

struct BlockState {
  // block barrier
  int bar_count = 0;
  unsigned bar_gen = 0;
  int nthreads = 0;
  // wave rendezvous
  int wave_count[kMaxThreads / kWave];
  unsigned wave_gen[kMaxThreads / kWave];
  int wave_size[kMaxThreads / kWave];
  unsigned char* scratch = nullptr;  // [nwaves][64*64]
};

struct Worker {
  ucontext_t sched;
  std::vector<Fiber> fibers;
  unsigned char* stacks = nullptr;
  unsigned char* smem = nullptr;
  BlockState bs;
  int current = -1;
  const std::function<void()>* body = nullptr;
};

thread_local Worker* tw = nullptr;

void fiber_entry() {
  Worker* w = tw;
  Fiber& f = w->fibers[w->current];
  (*w->body)();
  f.done = true;
  swapcontext(&f.ctx, &w->sched);
}

void yield_to_sched() {
  Worker* w = tw;
  Fiber& f = w->fibers[w->current];
  swapcontext(&f.ctx, &w->sched);
}

void run_block(Worker* w, dim3 grid, dim3 block, unsigned bx, unsigned by, unsigned bz) {
  const int n = (int)(block.x * block.y * block.z);
  if (n > kMaxThreads) {
    fprintf(stderr, "pp_emu: block too large (%d)\n", n);
    abort();
  }
  BlockState& bs = w->bs;
  bs.bar_count = 0;
  bs.bar_gen = 0;
  bs.nthreads = n;
  const int nwaves = (n + kWave - 1) / kWave;
  for (int i = 0; i < nwaves; ++i) {
    bs.wave_count[i] = 0;
    bs.wave_gen[i] = 0;
    bs.wave_size[i] = (i == nwaves - 1) ? (n - i * kWave) : kWave;
  }
  if ((int)w->fibers.size() < n) w->fibers.resize(n);
  for (int t = 0; t < n; ++t) {
    Fiber& f = w->fibers[t];
    f.done = false;
    f.tctx.tid.x = t % block.x;
    f.tctx.tid.y = (t / block.x) % block.y;
    f.tctx.tid.z = t / (block.x * block.y);
    f.tctx.bid = {bx, by, bz};
    f.tctx.bdim = block;
    f.tctx.gdim = grid;
    f.tctx.linear_tid = t;
    f.tctx.lane = t % kWave;
    f.tctx.wave = t / kWave;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = w->stacks + (size_t)t * kStackBytes;
    f.ctx.uc_stack.ss_size = kStackBytes;
    f.ctx.uc_link = &w->sched;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
  }
  int remaining = n;
  long spins = 0;
  while (remaining > 0) {
    int progressed = 0;
    for (int t = 0; t < n; ++t) {
      Fiber& f = w->fibers[t];
      if (f.done) continue;
      w->current = t;
      cur = &f.tctx;
      swapcontext(&w->sched, &f.ctx);
      if (f.done) {
        --remaining;
        ++progressed;
      }
    }
    if (++spins > 200000000L) {
      fprintf(stderr, "pp_emu: deadlock suspected (divergent barrier?)\n");
      abort();
    }
  }
  cur = nullptr;
}

}  // namespace

void barrier() {
  Worker* w = tw;
  BlockState& bs = w->bs;
  const unsigned gen = bs.bar_gen;
  if (++bs.bar_count == bs.nthreads) {
    bs.bar_count = 0;
    ++bs.bar_gen;
    return;
  }
  while (bs.bar_gen == gen) yield_to_sched();
}

void wave_sync() {
  Worker* w = tw;
  BlockState& bs = w->bs;
  const int wv = cur->wave;
  const unsigned gen = bs.wave_gen[wv];
  if (++bs.wave_count[wv] == bs.wave_size[wv]) {
    bs.wave_count[wv] = 0;
    ++bs.wave_gen[wv];
    return;
  }
  while (bs.wave_gen[wv] == gen) yield_to_sched();
}

unsigned char* wave_scratch() { return tw->bs.scratch + (size_t)cur->wave * kWave * 64; }

int active_lanes_in_wave() { return tw->bs.wave_size[cur->wave]; }

void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
  const long nblocks = (long)grid.x * grid.y * grid.z;
  if (nblocks <= 0) return;
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 4;
  const char* env = getenv("PP_EMU_THREADS");
  if (env) hw = (unsigned)atoi(env);
  int nworkers = (int)std::min<long>(nblocks, (long)hw);
  if (nworkers < 1) nworkers = 1;
  std::atomic<long> next(0);
  auto work = [&]() {
    Worker w;
    const size_t stack_total = (size_t)kMaxThreads * kStackBytes;
    w.stacks = (unsigned char*)mmap(nullptr, stack_total, PROT_READ | PROT_WRITE,
                                    MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (w.stacks == (unsigned char*)MAP_FAILED) {
      fprintf(stderr, "pp_emu: mmap failed\n");
      abort();
    }
    // dynamic LDS: the kernel may touch smem_bytes only; everything behind it is a canary checked after the last block (a
    // work-group writing past its launch's allocation would corrupt its neighbour's LDS on the GPU)
    constexpr size_t kLdsBytes = 160 * 1024, kCanaryTail = 4096;
    w.smem = (unsigned char*)aligned_alloc(64, kLdsBytes + kCanaryTail);
    const size_t used = std::min(smem_bytes, kLdsBytes);
    memset(w.smem + used, 0xA5, kLdsBytes + kCanaryTail - used);
    w.bs.scratch = (unsigned char*)aligned_alloc(64, (kMaxThreads / kWave) * kWave * 64);
    w.body = &body;
    tw = &w;
    dyn_smem = w.smem;
    for (;;) {
      long b = next.fetch_add(1);
      if (b >= nblocks) break;
      unsigned bx = (unsigned)(b % grid.x);
      unsigned by = (unsigned)((b / grid.x) % grid.y);
      unsigned bz = (unsigned)(b / ((long)grid.x * grid.y));
      run_block(&w, grid, block, bx, by, bz);
    }
    tw = nullptr;
    dyn_smem = nullptr;
    for (size_t i = used; i < kLdsBytes + kCanaryTail; ++i)
      if (w.smem[i] != 0xA5) {
        fprintf(stderr, "pp_emu: a kernel wrote dynamic LDS byte %zu, past the %zu bytes of its launch\n", i, smem_bytes);
        abort();
      }
    munmap(w.stacks, stack_total);
    free(w.smem);
    free(w.bs.scratch);
  };
  if (nworkers == 1) {
    // still run on a fresh thread: keeps the fibers' TLS separate from the caller
    std::thread t(work);
    t.join();
  } else {
    std::vector<std::thread> ts;
    for (int i = 0; i < nworkers; ++i) ts.emplace_back(work);
    for (auto& t : ts) t.join();
  }
}

}  // namespace pp_emu
