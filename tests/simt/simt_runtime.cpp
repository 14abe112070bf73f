// tests/simt/simt_runtime.cpp — TEST INFRASTRUCTURE ONLY: fiber scheduler of the host-side SIMT emulator.
#include <hip/hip_runtime.h>
#include <thread>
#include <vector>
#include <mutex>

namespace simt {
thread_local Lane* cur = nullptr;
static thread_local void* sched_sp = nullptr;
static thread_local const std::function<void()>* g_body = nullptr;

extern "C" void simt_switch(void** from_sp, void* to_sp);
asm(R"(
.text
.globl simt_switch
.type simt_switch,@function
simt_switch:
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
.size simt_switch,.-simt_switch
)");

static void fiber_main()
{
    Lane* me = cur;
    (*g_body)();
    me->done = true;
    // leaving lanes stop participating in rendezvous
    Wave& w = me->g->waves[me->wave];
    w.alive--; me->g->alive--;
    if (w.alive > 0 && w.arrived == w.alive) { w.arrived = 0; w.gen++; }
    if (me->g->alive > 0 && me->g->arrived == me->g->alive) { me->g->arrived = 0; me->g->gen++; }
    simt_switch(&me->sp, sched_sp);
    abort();
}

void yield()
{
    Lane* me = cur;
    simt_switch(&me->sp, sched_sp);
}

static const size_t kStack = 256 * 1024;

static void run_group(Group& g, dim3e bid, dim3e grid, dim3e block, const std::function<void()>& body)
{
    int const n = (int)(block.x * block.y * block.z);
    g.nthreads = n; g.nwaves = (n + 63) / 64; g.arrived = 0; g.alive = n; g.gen = 0;
    memset(g.smem, 0xCD, g.smemBytes);                   // LDS is NOT zero on a GPU: poison it
    for (int w = 0; w < g.nwaves; w++) { g.waves[w].arrived = 0; g.waves[w].gen = 0; g.waves[w].alive = (w == g.nwaves - 1) ? n - 64 * w : 64; }
    g_body = &body;
    for (int t = 0; t < n; t++) {
        Lane& L = g.lanes[t];
        L.done = false; L.g = &g; L.lane = t & 63; L.wave = t >> 6;
        L.tid = dim3e{ (unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y) };
        L.bid = bid; L.bdim = block; L.gdim = grid;
        uintptr_t top = ((uintptr_t)L.stack + kStack) & ~(uintptr_t)15;
        void** sp = (void**)top;
        *--sp = nullptr;                       // alignment slot / fake return address
        *--sp = (void*)&fiber_main;            // `ret` target
        for (int r = 0; r < 6; r++) *--sp = nullptr;
        L.sp = sp;
    }
    int remaining = n;
    while (remaining) {
        remaining = 0;
        for (int t = 0; t < n; t++) {
            Lane& L = g.lanes[t];
            if (L.done) continue;
            cur = &L;
            simt_switch(&sched_sp, L.sp);
            if (!L.done) remaining++;
        }
    }
    cur = nullptr;
}

void launch(dim3e grid, dim3e block, size_t smemBytes, const std::function<void()>& body, int nOsThreads)
{
    size_t const nGroups = (size_t)grid.x * grid.y * grid.z;
    int const n = (int)(block.x * block.y * block.z);
    if (nOsThreads <= 0) { nOsThreads = (int)std::thread::hardware_concurrency(); if (nOsThreads < 1) nOsThreads = 1; }
    if ((size_t)nOsThreads > nGroups) nOsThreads = (int)nGroups;
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        Group g; g.smemBytes = smemBytes ? smemBytes : 16; g.smem = (unsigned char*)aligned_alloc(64, (g.smemBytes + 63) & ~(size_t)63);
        g.lanes = new Lane[n];
        for (int t = 0; t < n; t++) g.lanes[t].stack = malloc(kStack);
        for (;;) {
            size_t const b = next.fetch_add(1);
            if (b >= nGroups) break;
            dim3e bid{ (unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((size_t)grid.x * grid.y)) };
            run_group(g, bid, grid, block, body);
        }
        for (int t = 0; t < n; t++) free(g.lanes[t].stack);
        delete[] g.lanes; free(g.smem);
    };
    if (nOsThreads <= 1) { worker(); return; }
    std::vector<std::thread> th;
    for (int i = 0; i < nOsThreads; i++) th.emplace_back(worker);
    for (auto& t : th) t.join();
}
}  // namespace simt
