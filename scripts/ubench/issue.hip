// scripts/ubench/issue.hip — measurement helper (not product): instruction issue rates of one CU as a function of resident
// waves, for SALU-only, VALU-only and mixed dependent chains.  hipcc --offload-arch=gfx950 -O3 issue.hip -o issue && ./issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define REP 256
template <int MODE>
__global__ void __launch_bounds__(64) k(uint64_t* out, int iters, uint32_t seed)
{
    uint32_t s = seed, v = threadIdx.x + seed, s2 = seed * 3, v2 = v * 5;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { asm volatile(".rept 256\n s_add_u32 %0, %0, 0x1357\n .endr" : "+s"(s) : : "scc"); }
        if (MODE == 1) { asm volatile(".rept 256\n v_add_u32 %0, 0x1357, %0\n .endr" : "+v"(v)); }
        if (MODE == 2) { asm volatile(".rept 128\n s_add_u32 %0, %0, 0x1357\n v_add_u32 %1, 0x1357, %1\n .endr" : "+s"(s), "+v"(v) : : "scc"); }
        if (MODE == 3) { asm volatile(".rept 85\n s_add_u32 %0, %0, 0x1357\n v_add_u32 %1, 0x1357, %1\n v_add_u32 %2, 0x2468, %2\n .endr\n s_nop 0" : "+s"(s), "+v"(v), "+v"(v2) : : "scc"); }
        if (MODE == 4) { asm volatile(".rept 128\n v_add_u32 %0, 0x1357, %0\n v_add_u32 %1, 0x2468, %1\n .endr" : "+v"(v), "+v"(v2)); }
        if (MODE == 5) { asm volatile(".rept 128\n s_add_u32 %0, %0, 0x1357\n s_add_u32 %1, %1, 0x2468\n .endr" : "+s"(s), "+s"(s2) : : "scc"); }
        if (MODE == 6) { asm volatile(".rept 256\n v_lshlrev_b64 %0, 1, %0\n .endr" : "+v"(*(uint64_t*)&v)); }
        if (MODE == 7) { asm volatile(".rept 128\n v_cmp_eq_u32 vcc, %0, %1\n s_and_b64 %2, vcc, exec\n .endr" : : "v"(v), "v"(v2), "s"(*(uint64_t*)&s) : "vcc", "scc"); }
        if (MODE == 8) { asm volatile(".rept 128\n v_readlane_b32 %0, %1, 3\n s_nop 0\n v_add_u32 %1, %0, %1\n .endr" : "+s"(s), "+v"(v) : : "scc"); }
        if (MODE == 9) { asm volatile(".rept 128\n v_add_u32 %0, 0x1357, %0\n s_nop 0\n .endr" : "+v"(v)); }
        if (MODE == 10) { asm volatile(".rept 128\n v_cmp_eq_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc\n .endr" : "+v"(v) : "v"(v2) : "vcc", "scc"); }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (s == 0x12345 && v == 77 && s2 == 1 && v2 == 3) out[0] = 0;
}

template <int MODE> void run(const char* name, int perInstr)
{
    uint64_t* d; hipMalloc(&d, 65536 * 8);
    int const iters = 64;
    for (int wpc : {1, 4, 9, 12, 16, 32}) {
        int const blocks = 256 * wpc;
        // LDS padding to control waves per CU is not needed: one block = one wave, the dispatcher spreads them evenly
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 4, 1u);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 1u);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint64_t> h(blocks);
        hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
        double sum = 0; for (auto x : h) sum += (double)x;
        double const ticks = sum / blocks;                 // s_memtime ticks (100 MHz constant clock on gfx9: convert by the caller)
        printf("%-30s waves/CU %2d  ticks/wave %9.0f  kernel %8.3f ms  ns per wave-instr %.3f (ticks %.3f)\n", name, wpc, ticks, ms, ms * 1e6 / ((double)iters * perInstr), ticks / ((double)iters * perInstr));
    }
    hipFree(d);
}

int main()
{
    run<0>("SALU dependent chain", 256);
    run<1>("VALU dependent chain", 256);
    run<2>("SALU+VALU 1:1", 256);
    run<3>("SALU+2VALU", 256);
    run<4>("2 VALU chains", 256);
    run<5>("2 SALU chains", 256);
    run<6>("v_lshlrev_b64 chain", 256);
    run<7>("v_cmp->s_and", 256);
    run<8>("v_readlane,nop,v_add dep", 384);
    run<9>("v_add,s_nop", 256);
    run<10>("v_cmp,nop1,v_cndmask dep", 384);
    return 0;
}
