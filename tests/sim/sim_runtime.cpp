/* tests/sim/sim_runtime.cpp -- fiber scheduler of the test-only SIMT emulator (see hip/hip_runtime.h). */
#include <hip/hip_runtime.h>
#include <sys/time.h>
#include <vector>

sim_uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;
unsigned char *sim_dynamic_shared = nullptr;

/* Fiber switching.  glibc's swapcontext makes a sigprocmask system call on every switch (~1 us); a kernel with wave
 * shuffles switches fibers hundreds of millions of times.  On x86-64 (and outside sanitizer builds, which need the
 * ucontext interceptors to follow the stacks) a fiber switch is therefore a hand-written save / restore of the
 * callee-saved registers: ~20x faster emulation. */
#if defined(__x86_64__) && !defined(__SANITIZE_ADDRESS__) && !defined(SIM_USE_UCONTEXT)
#define SIM_FAST_SWITCH 1
extern "C" void sim_switch(void **from_sp, void *to_sp);
__asm__(".text\n"
        ".globl sim_switch\n"
        ".type sim_switch,@function\n"
        "sim_switch:\n"
        "    pushq %rbp\n    pushq %rbx\n    pushq %r12\n    pushq %r13\n    pushq %r14\n    pushq %r15\n"
        "    movq %rsp, (%rdi)\n"
        "    movq %rsi, %rsp\n"
        "    popq %r15\n    popq %r14\n    popq %r13\n    popq %r12\n    popq %rbx\n    popq %rbp\n"
        "    ret\n"
        ".size sim_switch,.-sim_switch\n");
#else
#define SIM_FAST_SWITCH 0
#endif

namespace {
enum { RUN = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
struct Fiber {
#if SIM_FAST_SWITCH
    void *sp;
#else
    ucontext_t ctx;
#endif
    int state;
    unsigned tid;
    int site; /* source line of the barrier / cross-lane operation the fiber waits in (deadlock report) */
    int hist[8]; /* the sites before it, newest first */
};
const size_t kStack = 256 * 1024;
std::vector<Fiber> fibers;
std::vector<unsigned char> stacks;
#if SIM_FAST_SWITCH
void *sched_sp;
#else
ucontext_t sched_ctx;
#endif
int cur = -1;
const std::function<void()> *cur_body = nullptr;
uint64_t wave_slots[16][64];

void set_tid(unsigned t)
{
    threadIdx.x = t % blockDim.x;
    threadIdx.y = (t / blockDim.x) % blockDim.y;
    threadIdx.z = t / (blockDim.x * blockDim.y);
}
void to_scheduler(int me)
{
#if SIM_FAST_SWITCH
    sim_switch(&fibers[me].sp, sched_sp);
#else
    swapcontext(&fibers[me].ctx, &sched_ctx);
#endif
}
void to_fiber(unsigned t)
{
#if SIM_FAST_SWITCH
    sim_switch(&sched_sp, fibers[t].sp);
#else
    swapcontext(&sched_ctx, &fibers[t].ctx);
#endif
}
void fiber_main()
{
    (*cur_body)();
    fibers[cur].state = DONE;
    to_scheduler(cur);
    abort(); /* a finished fiber is never resumed */
}
void fiber_init(unsigned t)
{
    Fiber &f = fibers[t];
    f.state = RUN;
    f.tid = t;
    unsigned char *top = stacks.data() + (size_t)(t + 1) * kStack;
#if SIM_FAST_SWITCH
    uintptr_t T = (uintptr_t)top & ~(uintptr_t)15;
    void **sp = (void **)T;
    *--sp = nullptr;                 /* keeps the entry frame 16-byte aligned as after a call */
    *--sp = (void *)fiber_main;      /* sim_switch returns into the fiber's body */
    for (int i = 0; i < 6; i++) *--sp = nullptr; /* rbp rbx r12 r13 r14 r15 */
    f.sp = (void *)sp;
#else
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = stacks.data() + (size_t)t * kStack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = &sched_ctx;
    makecontext(&f.ctx, (void (*)())fiber_main, 0);
#endif
    (void)top;
}
void yield_as(int st)
{
    int me = cur;
    fibers[me].state = st;
    to_scheduler(me);
    set_tid(fibers[me].tid);
}
}  // namespace

void sim_set_site(int line)
{
    Fiber &f = fibers[cur];
    for (int i = 7; i > 0; i--) f.hist[i] = f.hist[i - 1];
    f.hist[0] = f.site;
    f.site = line;
}
void sim_block_barrier() { yield_as(WAIT_BLOCK); }
void sim_wave_barrier() { yield_as(WAIT_WAVE); }
uint64_t *sim_wave_slots() { return wave_slots[fibers[cur].tid / 64]; }
int sim_lane() { return (int)(fibers[cur].tid % 64); }
double sim_now()
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return tv.tv_sec + tv.tv_usec * 1e-6;
}

void sim_launch(const std::function<void()> &body, dim3 grid, dim3 block, size_t shmem)
{
    unsigned nthreads = block.x * block.y * block.z;
    if (nthreads == 0 || nthreads > 1024) { fprintf(stderr, "sim: bad block size %u\n", nthreads); abort(); }
    blockDim = block;
    gridDim = grid;
    std::vector<unsigned char> dyn(shmem + 64);
    sim_dynamic_shared = dyn.data();
    if (fibers.size() < nthreads) fibers.resize(nthreads);
    if (stacks.size() < (size_t)nthreads * kStack) stacks.resize((size_t)nthreads * kStack);
    cur_body = &body;
    unsigned nwaves = (nthreads + 63) / 64;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz;
                for (unsigned t = 0; t < nthreads; t++) fiber_init(t);
                for (;;) {
                    unsigned done = 0;
                    /* run every runnable fiber until it blocks */
                    for (unsigned t = 0; t < nthreads; t++) {
                        if (fibers[t].state == RUN) {
                            cur = (int)t;
                            set_tid(t);
                            to_fiber(t);
                        }
                    }
                    /* release wave barriers whose live lanes have all arrived */
                    bool progressed = false;
                    for (unsigned w = 0; w < nwaves; w++) {
                        unsigned lo = w * 64, hi = lo + 64 < nthreads ? lo + 64 : nthreads;
                        unsigned waiting = 0, live = 0;
                        for (unsigned t = lo; t < hi; t++) {
                            if (fibers[t].state != DONE) live++;
                            if (fibers[t].state == WAIT_WAVE) waiting++;
                        }
                        if (live && waiting == live) {
                            for (unsigned t = lo; t < hi; t++)
                                if (fibers[t].state == WAIT_WAVE) fibers[t].state = RUN;
                            progressed = true;
                        }
                    }
                    if (progressed) continue;
                    unsigned waiting = 0, live = 0;
                    for (unsigned t = 0; t < nthreads; t++) {
                        if (fibers[t].state == DONE) done++;
                        else live++;
                        if (fibers[t].state == WAIT_BLOCK) waiting++;
                    }
                    if (done == nthreads) break;
                    if (waiting == live) {
                        for (unsigned t = 0; t < nthreads; t++)
                            if (fibers[t].state == WAIT_BLOCK) fibers[t].state = RUN;
                        continue;
                    }
                    for (unsigned t = 0; t < nthreads; t++) /* where the minority waits: the divergent path */
                        if (fibers[t].state != DONE && (t % 64 == 0 || fibers[t].site != fibers[t - 1].site || fibers[t].state != fibers[t - 1].state))
                            fprintf(stderr, "sim:   thread %u.. waits at %s, source line %d (before: %d %d %d %d %d %d)\n", t, fibers[t].state == WAIT_BLOCK ? "the block barrier" : "a wave barrier / cross-lane operation", fibers[t].site,
                                    fibers[t].hist[0], fibers[t].hist[1], fibers[t].hist[2], fibers[t].hist[3], fibers[t].hist[4], fibers[t].hist[5]);
                    fprintf(stderr, "sim: deadlock in block (%u,%u,%u): %u live, %u at block barrier, rest at wave barriers\n",
                            bx, by, bz, live, waiting);
                    abort();
                }
            }
    sim_dynamic_shared = nullptr;
}
