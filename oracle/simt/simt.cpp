// Runtime of the functional SIMT simulator  --  TEST INFRASTRUCTURE ONLY (see stub/hip/hip_runtime.h).
//
// One fiber per GPU thread, scheduled cooperatively on the calling OS thread: a fiber runs until it reaches a wave
// collective or a workgroup barrier that is not yet complete, then yields to the scheduler; the last lane to arrive
// releases the others.  Workgroups of a launch run one after the other (LDS = `static` arrays persists between them,
// like on the hardware).  Context switches are a hand-written x86-64 callee-saved-register swap.
#include <sys/mman.h>

#include <hip/hip_runtime.h>

#undef asm
#undef volatile

#ifdef SIMT_POISON
extern "C" char __start_simt_lds[], __stop_simt_lds[];  // bounds of the section all LDS arrays live in (GNU ld)
#endif

namespace simt {

thread_local Fiber* cur = nullptr;

namespace {
thread_local void* g_sched_sp = nullptr;
const std::function<void()>* g_body = nullptr;  // shared, read-only during a launch
constexpr size_t STACK_BYTES = 512 * 1024;

extern "C" void simt_switch(void** save_sp, void* load_sp);
__asm__(
    ".text\n"
    ".globl simt_switch\n"
    ".type simt_switch,@function\n"
    "simt_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n");

void release_if_complete(Fiber* f) {
  Wave* w = f->wave;
  if (w->live > 0 && w->arrived == w->live) { w->arrived = 0; w->gen++; }
  Block* b = f->block;
  if (b->live > 0 && b->arrived == b->live) { b->arrived = 0; b->gen++; }
}

extern "C" void simt_fiber_main() {
  Fiber* f = cur;
  (*g_body)();
  f->done = true;
  f->wave->live--;
  f->block->live--;
  release_if_complete(f);  // lanes / waves still waiting must not wait for one that has exited
  simt_switch(&f->sp, g_sched_sp);
  std::abort();  // a finished fiber is never resumed
}

void yield() {
  Fiber* f = cur;
  simt_switch(&f->sp, g_sched_sp);
}

struct StackPool {
  ~StackPool() { for (void* p : free_) munmap(p, STACK_BYTES); }
  std::vector<void*> free_;
  void* get() {
    if (!free_.empty()) { void* p = free_.back(); free_.pop_back(); return p; }
    void* p = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { std::perror("simt: mmap"); std::abort(); }
    return p;
  }
  void put(void* p) { free_.push_back(p); }
};
thread_local StackPool g_stacks;
}  // namespace

void wave_sync() {
  Fiber* f = cur;
  Wave* w = f->wave;
  const unsigned my = w->gen;
  if (++w->arrived == w->live) { w->arrived = 0; w->gen++; return; }
  f->waiting = 1;
  f->wgen = my;
  while (w->gen == my) yield();
  f->waiting = 0;
}

void block_sync() {
  Fiber* f = cur;
  Block* b = f->block;
  const unsigned my = b->gen;
  if (++b->arrived == b->live) { b->arrived = 0; b->gen++; return; }
  f->waiting = 2;
  f->bgen = my;
  while (b->gen == my) yield();
  f->waiting = 0;
}

bool dma_late() {
  static const bool late = [] { const char* e = std::getenv("SAMAUDIO_SIMT_DMA"); return e && std::string(e) == "late"; }();
  return late;
}

void wait_vmcnt(int n) {
  wave_sync();
  Wave* w = cur->wave;  // the first lane through does the work, the others find nothing left to do
  while ((long)w->pending.size() > (long)n) {
    Wave::Dma& d = w->pending.front();
    for (int l = 0; l < WAVE; ++l) std::memcpy(d.base + 16 * l, d.data[l], 16);
    w->pending.pop_front();
    w->dma_first++;
  }
}

thread_local bool dma_asm = false;
void wait_vmcnt_if_visible() {
  wave_sync();
  Wave* w = cur->wave;
  bool any = false;
  for (const Wave::Dma& d : w->pending) any = any || d.visible;
  if (any) wait_vmcnt(0);
  else wave_sync();   // same number of wave-synchronous points on both paths
}

// `asm volatile("...")` statements of the kernels: a wave-synchronous point; `s_waitcnt vmcnt(<literal>)` retires DMAs.
// (The one statement whose count is an asm operand, gemm2.hip wait_vmcnt<N>, is rewritten to simt::wait_vmcnt(N) by
// oracle/simt/build.sh - the preprocessor cannot see an operand's value.)
void asm_stmt(const char* text) {
  if (std::strstr(text, "s_nop")) return;   // idle cycles: no meaning here, and legal under divergent control flow (no wave sync)
  const char* v = std::strstr(text, "vmcnt(");
  if (v) {
    if (v[6] < '0' || v[6] > '9') { std::fprintf(stderr, "simt: vmcnt with a non-literal count: %s\n", text); std::abort(); }
    wait_vmcnt(std::atoi(v + 6));
    return;
  }
  wave_sync();
}

const unsigned char* row_publish(const void* data, size_t n) {
  Fiber* f = cur;
  Wave* w = f->wave;
  const int row = f->lane >> 4;
  if (n > 8) { std::fprintf(stderr, "simt: row publish of %zu bytes\n", n); std::abort(); }
  unsigned char* buf = &w->rbuf[f->rcount & 1][0][0];
  std::memcpy(buf + (size_t)f->lane * 8, data, n);
  f->rcount++;
  const unsigned my = w->row_gen[row];
  if (++w->row_arrived[row] == 16) { w->row_arrived[row] = 0; w->row_gen[row]++; return buf; }
  f->waiting = 3;
  f->rgen = my;
  while (w->row_gen[row] == my) yield();
  f->waiting = 0;
  return buf;
}

const unsigned char* wave_publish(const void* data, size_t n) {
  Fiber* f = cur;
  if (n > (size_t)XBYTES) { std::fprintf(stderr, "simt: publish of %zu bytes\n", n); std::abort(); }
  unsigned char* buf = &f->wave->xbuf[f->xcount & 1][0][0];
  std::memcpy(buf + (size_t)f->lane * XBYTES, data, n);
  f->xcount++;
  wave_sync();
  return buf;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  if (cur) { std::fprintf(stderr, "simt: nested launch\n"); std::abort(); }
  const int nthreads = (int)(block.x * block.y * block.z);
  const int nwaves = (nthreads + WAVE - 1) / WAVE;
  static const bool reverse = [] { const char* e = std::getenv("SAMAUDIO_SIMT_ORDER"); return e && std::string(e) == "reverse"; }();
  g_body = &body;
  const long nblocks = (long)grid.x * grid.y * grid.z;
  // Workgroups are independent (no kernel of this library communicates between workgroups), so they are simulated
  // concurrently: one OS thread = one scheduler + its own LDS image (thread_local `static` arrays) + its own fibers.
#ifdef SIMT_POISON
#pragma omp parallel num_threads(1)
#else
#pragma omp parallel
#endif
  {
    std::vector<Fiber> fibers((size_t)nthreads);
    std::vector<Wave> waves((size_t)nwaves);
    for (int i = 0; i < nthreads; ++i) fibers[i].stack = g_stacks.get();
#pragma omp for schedule(dynamic, 1)
    for (long blk_id = 0; blk_id < nblocks; ++blk_id) {
      const unsigned bx = (unsigned)(blk_id % grid.x), by = (unsigned)((blk_id / grid.x) % grid.y),
                     bz = (unsigned)(blk_id / ((long)grid.x * grid.y));
#ifdef SIMT_POISON
      // SAMAUDIO_SIMT_POISON_BYTE: the fill byte (default 0xFF = NaN; a finite pattern such as 0x3F shows reads of unwritten
      // LDS that NaN-ignoring operations - fmaxf, selects - hide)
      static const int fill = [] { const char* e = std::getenv("SAMAUDIO_SIMT_POISON_BYTE"); return e ? (int)std::strtol(e, nullptr, 0) : 0xFF; }();
      std::memset(__start_simt_lds, fill, (size_t)(__stop_simt_lds - __start_simt_lds));
#endif
      Block blk;
      blk.live = nthreads;
      for (int w = 0; w < nwaves; ++w) {
        waves[w].live = std::min(WAVE, nthreads - w * WAVE);
        waves[w].arrived = 0;
        waves[w].gen = 0;
        for (int r = 0; r < 4; ++r) { waves[w].row_arrived[r] = 0; waves[w].row_gen[r] = 0; }
        waves[w].pending.clear();  // DMAs still in flight when a workgroup ends are dropped
        waves[w].dma_first = 0;
      }
      for (int i = 0; i < nthreads; ++i) {
        Fiber& f = fibers[i];
        void* stack = f.stack;
        f = Fiber{};
        f.stack = stack;
        f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
        f.bid = dim3(bx, by, bz);
        f.bdim = block;
        f.gdim = grid;
        f.lane = i % WAVE;
        f.wave = &waves[i / WAVE];
        f.block = &blk;
        // initial frame: six callee-saved slots, then the entry address; rsp is 16-byte aligned + 8 at entry
        uintptr_t top = ((uintptr_t)stack + STACK_BYTES) & ~(uintptr_t)15;
        void** sp = (void**)(top - 8);
        *--sp = (void*)&simt_fiber_main;
        for (int k = 0; k < 6; ++k) *--sp = nullptr;
        f.sp = sp;
      }
      int remaining = nthreads;
      while (remaining > 0) {
        bool progress = false;
        for (int ii = 0; ii < nwaves * WAVE; ++ii) {
          // SAMAUDIO_SIMT_ORDER=reverse runs the waves of a workgroup in the opposite order between barriers, so that
          // a hazard between two waves inside one barrier interval is seen from both sides (with the early DMA mode: a
          // wave staging over data another wave of the same interval still has to read)
          const int i = reverse ? (nwaves - 1 - ii / WAVE) * WAVE + ii % WAVE : ii;
          if (i >= nthreads) continue;
          Fiber& f = fibers[i];
          if (f.done) continue;
          if (f.waiting == 1 && f.wave->gen == f.wgen) continue;
          if (f.waiting == 2 && f.block->gen == f.bgen) continue;
          if (f.waiting == 3 && f.wave->row_gen[f.lane >> 4] == f.rgen) continue;
          cur = &f;
          simt_switch(&g_sched_sp, f.sp);
          cur = nullptr;
          progress = true;
          if (f.done) --remaining;
        }
        if (!progress) {
          std::fprintf(stderr, "simt: deadlock in block (%u,%u,%u): %d threads wait at a barrier / collective the "
                               "others never reach\n", bx, by, bz, remaining);
          std::abort();
        }
      }
    }
    for (int i = 0; i < nthreads; ++i) g_stacks.put(fibers[i].stack);
  }
  g_body = nullptr;
}

}  // namespace simt
