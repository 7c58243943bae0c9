// Fiber scheduler behind tests/emu/hip_emu.h.  TEST INFRASTRUCTURE (see the header).
#include "hip_emu.h"

#include <ucontext.h>

#include <vector>

namespace emu {
dim3 threadIdx_, blockIdx_, blockDim_, gridDim_;
char* dyn_smem = nullptr;

namespace {
enum State { RUN, BAR, WAVE, DONE };
struct Fiber {
    ucontext_t ctx;
    std::vector<char> stack;
    State st = RUN;
    unsigned tx = 0, ty = 0, tz = 0;
    int lin = 0;
    unsigned bar_gen = 0;    // generation this fiber waits to be exceeded
    unsigned wave_seq = 0;   // number of wave exchanges this lane has entered
};
constexpr size_t kStack = 256 * 1024;
ucontext_t sched_ctx;
std::vector<Fiber> fibers;
Fiber* cur = nullptr;
const std::function<void()>* body_fn = nullptr;
unsigned bar_generation = 0, bar_arrived = 0, n_alive = 0;
struct WaveBuf {
    char slot[2][64][kSlot];
    unsigned arrived[2];   // lanes arrived for parity p
    unsigned seq_open[2];  // which seq the parity buffer currently belongs to
};
std::vector<WaveBuf> waves;
std::vector<unsigned> wave_lanes;  // live lanes per wave

void trampoline() {
    (*body_fn)();
    cur->st = DONE;
    --n_alive;
    --wave_lanes[cur->lin / 64];
    swapcontext(&cur->ctx, &sched_ctx);
}

void yield_to_sched() { swapcontext(&cur->ctx, &sched_ctx); }

}  // namespace

int lane_id() { return cur->lin & 63; }

void syncthreads() {
    cur->bar_gen = bar_generation;
    cur->st = BAR;
    if (++bar_arrived >= n_alive) {
        bar_arrived = 0;
        ++bar_generation;
    }
    yield_to_sched();
    cur->st = RUN;
}

const char* wave_exchange(const void* payload, int bytes) {
    if (bytes > kSlot) { fprintf(stderr, "emu: payload too large\n"); abort(); }
    WaveBuf& w = waves[cur->lin / 64];
    unsigned seq = cur->wave_seq++;
    unsigned p = seq & 1;
    if (w.seq_open[p] != seq) {  // first lane to arrive for this exchange: recycle the buffer
        w.seq_open[p] = seq;
        w.arrived[p] = 0;
    }
    memcpy(w.slot[p][cur->lin & 63], payload, bytes);
    ++w.arrived[p];
    cur->st = WAVE;
    while (w.arrived[p] < wave_lanes[cur->lin / 64]) yield_to_sched();
    cur->st = RUN;
    return &w.slot[p][0][0];
}

void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
    const unsigned nthreads = block.x * block.y * block.z;
    if (fibers.size() < nthreads) {
        fibers.resize(nthreads);
        for (auto& f : fibers)
            if (f.stack.empty()) f.stack.resize(kStack);
    }
    std::vector<char> smem(smem_bytes + 64);
    dyn_smem = (char*)(((uintptr_t)smem.data() + 63) & ~(uintptr_t)63);
    gridDim_ = grid;
    blockDim_ = block;
    body_fn = &body;
    const unsigned nwaves = (nthreads + 63) / 64;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                waves.assign(nwaves, WaveBuf{});
                for (auto& w : waves) w.seq_open[0] = w.seq_open[1] = 0xffffffffu;
                wave_lanes.assign(nwaves, 0);
                bar_generation = bar_arrived = 0;
                n_alive = nthreads;
                for (unsigned t = 0; t < nthreads; ++t) {
                    Fiber& f = fibers[t];
                    f.st = RUN;
                    f.lin = (int)t;
                    f.tx = t % block.x;
                    f.ty = (t / block.x) % block.y;
                    f.tz = t / (block.x * block.y);
                    f.bar_gen = 0;
                    f.wave_seq = 0;
                    ++wave_lanes[t / 64];
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack.data();
                    f.ctx.uc_stack.ss_size = f.stack.size();
                    f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, trampoline, 0);
                }
                while (n_alive > 0) {
                    bool progress = false;
                    for (unsigned t = 0; t < nthreads; ++t) {
                        Fiber& f = fibers[t];
                        if (f.st == DONE) continue;
                        bool ok = f.st == RUN || (f.st == BAR && bar_generation != f.bar_gen) ||
                                  (f.st == WAVE);
                        if (f.st == WAVE) {
                            WaveBuf& w = waves[t / 64];
                            unsigned p = (f.wave_seq - 1) & 1;
                            ok = w.arrived[p] >= wave_lanes[t / 64];
                        }
                        if (!ok) continue;
                        cur = &f;
                        threadIdx_ = dim3(f.tx, f.ty, f.tz);
                        blockIdx_ = dim3(bx, by, bz);
                        swapcontext(&sched_ctx, &f.ctx);
                        progress = true;
                    }
                    if (!progress) {
                        fprintf(stderr, "emu: deadlock in block (%u,%u,%u): divergent barrier?\n", bx, by, bz);
                        abort();
                    }
                }
            }
    dyn_smem = nullptr;
}
}  // namespace emu
