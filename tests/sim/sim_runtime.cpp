/* tests/sim/sim_runtime.cpp -- fiber scheduler of the test-only SIMT emulator (see hip/hip_runtime.h). */
#include <hip/hip_runtime.h>
#include <sys/time.h>
#include <vector>

sim_uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;
unsigned char *sim_dynamic_shared = nullptr;

namespace {
enum { RUN = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
struct Fiber {
    ucontext_t ctx;
    int state;
    unsigned tid;
};
const size_t kStack = 256 * 1024;
std::vector<Fiber> fibers;
std::vector<unsigned char> stacks;
ucontext_t sched_ctx;
int cur = -1;
const std::function<void()> *cur_body = nullptr;
uint64_t wave_slots[16][64];

void set_tid(unsigned t)
{
    threadIdx.x = t % blockDim.x;
    threadIdx.y = (t / blockDim.x) % blockDim.y;
    threadIdx.z = t / (blockDim.x * blockDim.y);
}
void fiber_main()
{
    (*cur_body)();
    fibers[cur].state = DONE;
    swapcontext(&fibers[cur].ctx, &sched_ctx);
}
void yield_as(int st)
{
    int me = cur;
    fibers[me].state = st;
    swapcontext(&fibers[me].ctx, &sched_ctx);
    set_tid(fibers[me].tid);
}
}  // namespace

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
                for (unsigned t = 0; t < nthreads; t++) {
                    Fiber &f = fibers[t];
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = stacks.data() + (size_t)t * kStack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = &sched_ctx;
                    f.state = RUN;
                    f.tid = t;
                    makecontext(&f.ctx, (void (*)())fiber_main, 0);
                }
                for (;;) {
                    unsigned done = 0;
                    /* run every runnable fiber until it blocks */
                    for (unsigned t = 0; t < nthreads; t++) {
                        if (fibers[t].state == RUN) {
                            cur = (int)t;
                            set_tid(t);
                            swapcontext(&sched_ctx, &fibers[t].ctx);
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
                    fprintf(stderr, "sim: deadlock in block (%u,%u,%u): %u live, %u at block barrier, rest at wave barriers\n",
                            bx, by, bz, live, waiting);
                    abort();
                }
            }
    sim_dynamic_shared = nullptr;
}
