// hipemu scheduler: one fiber per GPU thread (hand-rolled x86-64 context switch: glibc's swapcontext makes a sigprocmask
// system call per switch, which dominated the run time of barrier-heavy kernels), blocks distributed over host threads.
// TEST INFRASTRUCTURE ONLY - see hip/hip_runtime.h in this directory.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <mutex>

namespace hipemu {
thread_local BlockRunner* g_runner = nullptr;
thread_local uint3 g_tid, g_bid;
thread_local dim3 g_bdim, g_gdim;

static const size_t kStack = 96 * 1024;

char* dyn_smem() { return g_runner->dyn; }

#if !defined(__x86_64__)
#error "hipemu's context switch is written for x86-64 (System V ABI)"
#endif
// void hipemu_switch(void** save_sp, void* new_sp): push the callee-saved registers, publish the stack pointer, adopt the
// other one, pop its registers, return into it.  (MXCSR / x87 control words are never changed by the emulated code.)
extern "C" void hipemu_switch(void** save_sp, void* new_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
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
.size hipemu_switch, .-hipemu_switch
)");

void yield_state(int st) {
  BlockRunner* r = g_runner;
  Fiber& f = r->fibers[r->cur];
  f.state = st;
  hipemu_switch(&f.sp, r->sched_sp);
}

static void fiber_entry() {
  BlockRunner* r = g_runner;
  r->body();
  r->fibers[r->cur].state = DONE;
  hipemu_switch(&r->fibers[r->cur].sp, r->sched_sp);
  __builtin_trap();   // a finished fiber is never resumed
}

static void run_block(BlockRunner* r, dim3 block, size_t shmem) {
  int n = (int)(block.x * block.y * block.z);
  r->nthreads = n;
  if ((int)r->fibers.size() < n) {
    size_t old = r->fibers.size();
    r->fibers.resize(n);
    for (size_t i = old; i < (size_t)n; ++i) r->fibers[i].stack = (char*)malloc(kStack);
  }
  r->xa.assign(((n + 63) / 64) * 64, 0);
  r->xb.assign(((n + 63) / 64) * 64, 0);
  r->xw.assign(((n + 63) / 64) * 64 * 8, 0);
  if (shmem > r->dyn_cap) {
    free(r->dyn);
    r->dyn = (char*)aligned_alloc(64, (shmem + 63) / 64 * 64);
    r->dyn_cap = shmem;
  }
  for (int i = 0; i < n; ++i) {
    Fiber& f = r->fibers[i];
    // initial frame: six zeroed callee-saved registers, then fiber_entry as the return address; after that `ret` the stack
    // pointer is 8 (mod 16), exactly as after a call instruction
    uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStack) & ~(uintptr_t)15;
    void** sp = reinterpret_cast<void**>(top);
    *--sp = nullptr;                                   // alignment slot
    *--sp = reinterpret_cast<void*>(&fiber_entry);     // return address of the first switch
    for (int k = 0; k < 6; ++k) *--sp = nullptr;       // rbp rbx r12 r13 r14 r15
    f.sp = sp;
    f.state = READY;
  }
  int ndone = 0;
  while (ndone < n) {
    bool progressed = false;
    for (int i = 0; i < n; ++i) {
      Fiber& f = r->fibers[i];
      if (f.state != READY) continue;
      r->cur = i;
      g_tid.x = i % block.x;
      g_tid.y = (i / block.x) % block.y;
      g_tid.z = i / (block.x * block.y);
      hipemu_switch(&r->sched_sp, f.sp);
      progressed = true;
      if (f.state == DONE) ++ndone;
    }
    // release wave-level sync points
    bool released = false;
    for (int w = 0; w * 64 < n; ++w) {
      int lo = w * 64, hi = lo + 64 < n ? lo + 64 : n;
      int waiting = 0, alive = 0;
      for (int i = lo; i < hi; ++i) {
        if (r->fibers[i].state != DONE) ++alive;
        if (r->fibers[i].state == WAIT_WAVE) ++waiting;
      }
      if (alive && waiting == alive) {
        for (int i = lo; i < hi; ++i)
          if (r->fibers[i].state == WAIT_WAVE) r->fibers[i].state = READY;
        released = true;
      }
    }
    if (!released) {
      int waiting = 0, alive = 0;
      for (int i = 0; i < n; ++i) {
        if (r->fibers[i].state != DONE) ++alive;
        if (r->fibers[i].state == WAIT_BLOCK) ++waiting;
      }
      if (alive && waiting == alive) {
        for (int i = 0; i < n; ++i)
          if (r->fibers[i].state == WAIT_BLOCK) r->fibers[i].state = READY;
        released = true;
      }
    }
    if (!released && !progressed && ndone < n) {
      fprintf(stderr, "hipemu: DEADLOCK in block (%u,%u,%u): divergent barrier / wave op\n", g_bid.x, g_bid.y, g_bid.z);
      for (int i = 0; i < n; i += 1)
        if (r->fibers[i].state != DONE && (i % 32 == 0)) fprintf(stderr, "  tid %d state %d\n", i, r->fibers[i].state);
      abort();
    }
  }
}

// persistent worker pool (fresh std::threads per launch would re-allocate the module's TLS block,
// which holds every `__shared__` array, on each launch)
struct Pool {
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::vector<std::thread> threads;
  std::function<void()> job;
  uint64_t gen = 0;
  int pending = 0;
  bool stop = false;
  void ensure(size_t n) {
    while (threads.size() < n) {
      threads.emplace_back([this]() {
        uint64_t seen = 0;
        for (;;) {
          std::function<void()> j;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv_job.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            j = job;
          }
          j();
          {
            std::unique_lock<std::mutex> lk(mu);
            if (--pending == 0) cv_done.notify_all();
          }
        }
      });
    }
  }
  void run(const std::function<void()>& j) {
    std::unique_lock<std::mutex> lk(mu);
    job = j;
    pending = (int)threads.size();
    ++gen;
    cv_job.notify_all();
    cv_done.wait(lk, [&] { return pending == 0; });
  }
};
static Pool* g_pool = nullptr;
Graph*& capturing_graph() {
  static Graph* g = nullptr;
  return g;
}
static std::mutex g_launch_mu;

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  size_t nblocks = (size_t)grid.x * grid.y * grid.z;
  if (nblocks == 0) return;
  std::lock_guard<std::mutex> guard(g_launch_mu);
  unsigned hw = std::thread::hardware_concurrency();
  const char* env = getenv("HIPEMU_THREADS");
  if (env) hw = (unsigned)atoi(env);
  if (hw < 1) hw = 1;
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    static thread_local BlockRunner runner;
    BlockRunner* r = &runner;
    g_runner = r;
    r->body = body;
    g_bdim = block;
    g_gdim = grid;
    for (;;) {
      size_t b = next.fetch_add(1);
      if (b >= nblocks) break;
      g_bid.x = (unsigned)(b % grid.x);
      g_bid.y = (unsigned)((b / grid.x) % grid.y);
      g_bid.z = (unsigned)(b / ((size_t)grid.x * grid.y));
      run_block(r, block, shmem);
    }
  };
  if (hw <= 1 || nblocks == 1) {
    worker();
  } else {
    if (!g_pool) g_pool = new Pool();
    g_pool->ensure(hw);
    g_pool->run(worker);
  }
}
}  // namespace hipemu
