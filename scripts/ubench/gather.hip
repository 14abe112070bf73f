// scripts/ubench/gather.hip — measurement helper (not product): cost of the parse kernels' load shapes, per wave, dependent round trips.
// Each wave owns a 128 KB region (like a unit); 9 waves per CU resident (LDS padding), loads hit L2/MALL/HBM like the parser's do.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
struct Q { uint32_t x, y, z, w; };
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ Q ld128(const uint8_t* p) { Q v; __builtin_memcpy(&v, p, 16); return v; }

template <int MODE>
__global__ void __launch_bounds__(64) k(const uint8_t* src, uint64_t* out, int iters)
{
    extern __shared__ unsigned char smem[];
    const uint8_t* base = src + (size_t)blockIdx.x * 131072;
    uint32_t lane = threadIdx.x;
    uint32_t pos = 4096, acc = 0;
    uint32_t rnd = lane * 2654435761u + blockIdx.x * 40503u;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        rnd = rnd * 1664525u + 1013904223u;
        uint32_t cand = 64 + ((rnd >> 8) % (pos - 64));            // random earlier position (unaligned)
        uint32_t P = pos + lane;
        if (MODE == 0) { acc += ld32(base + cand); }                                         // 4 B gather
        if (MODE == 1) { Q q = ld128(base + cand - 4); acc += q.x ^ q.y ^ q.z ^ q.w; }      // 16 B gather, byte-aligned
        if (MODE == 2) { Q q = ld128(base + cand - 4), r = ld128(base + cand + 12); acc += q.x ^ q.y ^ q.z ^ q.w ^ r.x ^ r.y ^ r.z ^ r.w; }   // 32 B gather
        if (MODE == 3) { Q q = ld128(base + ((cand - 4) & ~3u)); acc += q.x ^ q.y ^ q.z ^ q.w; }    // 16 B gather, 4-aligned
        if (MODE == 4) { Q q = ld128(base + ((cand - 4) & ~15u)), r = ld128(base + ((cand - 4) & ~15u) + 16); acc += q.x ^ q.y ^ q.z ^ q.w ^ r.x ^ r.y ^ r.z ^ r.w; }   // 2 x 16 B, 16-aligned
        if (MODE == 5) { uint64_t v = ld64(base + P); acc += (uint32_t)v ^ (uint32_t)(v >> 32); }       // own bytes 8 B, byte stride
        if (MODE == 6) { Q q = ld128(base + P - 4), r = ld128(base + P + 12); acc += q.x ^ q.y ^ q.z ^ q.w ^ r.x ^ r.y ^ r.z ^ r.w; }    // own bytes 32 B, byte stride
        if (MODE == 7) { uint64_t v = ld64(base + ((P & ~63u) + 8 * (lane & 15))); acc += (uint32_t)v; }   // aligned coalesced 128 B
        if (MODE == 8) { uint64_t v = ld64(base + cand); acc += (uint32_t)v ^ (uint32_t)(v >> 32); }    // 8 B gather
        // make the next iteration depend on the data (a dependent round trip, like the parser's)
        pos += 60 + (__builtin_amdgcn_readfirstlane(acc) & 3);
        if (pos > 131072 - 256) pos = 4096;
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678) out[0] = smem[0];
}
template <int MODE> void run(const char* name, const uint8_t* src, uint64_t* d)
{
    int const iters = 2000, blocks = 8192;
    size_t const lds = 17408;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), lds, 0, src, d, 50);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), lds, 0, src, d, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(blocks);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto x : h) sum += (double)x;
    printf("%-44s kernel %8.3f ms   cycles per dependent round: %8.1f\n", name, ms, sum / blocks / iters);
}
int main()
{
    uint8_t* src; uint64_t* d;
    size_t const n = (size_t)8192 * 131072;
    hipMalloc(&src, n + 4096); hipMalloc(&d, 8192 * 8);
    hipMemset(src, 0x5a, n + 4096);
    run<0>("gather 4 B (today's candidate fetch)", src, d);
    run<8>("gather 8 B", src, d);
    run<1>("gather 16 B byte-aligned", src, d);
    run<3>("gather 16 B 4-aligned", src, d);
    run<2>("gather 2 x 16 B byte-aligned", src, d);
    run<4>("gather 2 x 16 B 16-aligned", src, d);
    run<5>("own bytes 8 B, byte stride (today)", src, d);
    run<6>("own bytes 2 x 16 B, byte stride", src, d);
    run<7>("own bytes aligned coalesced", src, d);
    return 0;
}
