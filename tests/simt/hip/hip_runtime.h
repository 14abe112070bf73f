// tests/simt/hip/hip_runtime.h — TEST INFRASTRUCTURE ONLY.
//
// A host-side SIMT *emulator* that stands in for <hip/hip_runtime.h> when the product's device headers
// (zstd_amd/csrc/zhip_*.h) are compiled with g++ for debugging on a machine without a GPU.  The product sources
// are plain gfx950 HIP with no #ifdefs; tests add `-I tests/simt` so that `#include <hip/hip_runtime.h>` resolves
// here.  Every work-item of a workgroup is a fiber; wave collectives (__ballot, __shfl, readlane ...) and
// __syncthreads() are rendezvous points, so the code between two collectives runs lane-by-lane in arbitrary order —
// which is stricter than the hardware's lockstep and flushes out missing barriers.  Nothing here ships.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <functional>
#include <atomic>

namespace simt {

struct dim3e { unsigned x, y, z; };

struct Lane;
struct Wave {
    uint64_t slot[64];      // contribution of each lane to the current collective
    uint64_t result;        // reduced result (ballot)
    int arrived, alive;
    unsigned gen;
    int kind; const void* site;   // sanity: all lanes must be in the same collective
};
struct Group {
    Wave waves[16];
    int nthreads, nwaves;
    int arrived, alive; unsigned gen;          // __syncthreads rendezvous
    unsigned char* smem; size_t smemBytes;
    Lane* lanes;
};
struct Lane {
    void* sp; void* stack; bool done;
    dim3e tid, bid, bdim, gdim;
    int lane, wave;
    Group* g;
};

extern thread_local Lane* cur;            // the running fiber
void yield();                              // switch to the next runnable fiber of the group
void launch(dim3e grid, dim3e block, size_t smemBytes, const std::function<void()>& body, int nOsThreads = 0);

enum { K_BALLOT = 1, K_SHFL, K_READLANE, K_FIRST, K_WBAR, K_ANY };

// rendezvous of the alive lanes of the current wave; phase A publishes, phase B lets everybody read
static inline void wave_rendezvous(int kind, const void* site)
{
    Lane* me = cur; Wave& w = me->g->waves[me->wave];
    if (w.arrived == 0) { w.kind = kind; w.site = site; }
    else if (w.kind != kind) { fprintf(stderr, "simt: divergent collectives in wave %d (kind %d vs %d)\n", me->wave, w.kind, kind); abort(); }
    unsigned const g0 = w.gen;
    if (++w.arrived == w.alive) { w.arrived = 0; w.gen++; }
    else while (w.gen == g0) yield();
}

template <typename T> static inline uint64_t bits_of(T v) { uint64_t b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template <typename T> static inline T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }

static inline uint64_t ballot_impl(bool p)
{
    Lane* me = cur; Wave& w = me->g->waves[me->wave];
    w.slot[me->lane] = p ? 1 : 0;
    wave_rendezvous(K_BALLOT, nullptr);
    uint64_t m = 0;
    for (int i = 0; i < 64; i++) {
        Lane* L = &me->g->lanes[me->wave * 64 + i];
        if (me->wave * 64 + i < me->g->nthreads && !L->done && w.slot[i]) m |= 1ull << i;
    }
    wave_rendezvous(K_BALLOT, nullptr);
    return m;
}
template <typename T> static inline T shfl_impl(T v, int srcLane)
{
    Lane* me = cur; Wave& w = me->g->waves[me->wave];
    w.slot[me->lane] = bits_of(v);
    wave_rendezvous(K_SHFL, nullptr);
    uint64_t const r = w.slot[srcLane & 63];
    wave_rendezvous(K_SHFL, nullptr);
    return from_bits<T>(r);
}
static inline int first_alive_lane()
{
    Lane* me = cur;
    for (int i = 0; i < 64; i++) if (me->wave * 64 + i < me->g->nthreads && !me->g->lanes[me->wave * 64 + i].done) return i;
    return 0;
}
}  // namespace simt

// ---------------------------------------------------------------- language surface used by zstd_amd/csrc
#define ZHIP_LDS                      /* LDS address-space qualifier of the product headers: plain memory here */
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or(p, v, order)
#define __hip_atomic_fetch_and(p, v, order, scope) __atomic_fetch_and(p, v, order)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add(p, v, order)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)simt::cur->g->smem;

#define threadIdx (simt::cur->tid)
#define blockIdx  (simt::cur->bid)
#define blockDim  (simt::cur->bdim)
#define gridDim   (simt::cur->gdim)

static inline unsigned long long __ballot(int p) { return simt::ballot_impl(p != 0); }
static inline int __any(int p) { return __ballot(p) != 0; }
static inline int __all(int p) { return __ballot(!p) == 0; }
template <typename T> static inline T __shfl(T v, int lane, int = 64) { return simt::shfl_impl(v, lane); }
template <typename T> static inline T __shfl_up(T v, unsigned d, int = 64)
{ int const l = simt::cur->lane; T r = simt::shfl_impl(v, l - (int)d < 0 ? l : l - (int)d); return r; }
template <typename T> static inline T __shfl_down(T v, unsigned d, int = 64)
{ int const l = simt::cur->lane; T r = simt::shfl_impl(v, l + (int)d > 63 ? l : l + (int)d); return r; }
template <typename T> static inline T __shfl_xor(T v, int m, int = 64) { return simt::shfl_impl(v, simt::cur->lane ^ m); }

static inline unsigned __builtin_amdgcn_readlane(unsigned v, int lane) { return simt::shfl_impl(v, lane); }
static inline unsigned __builtin_amdgcn_readfirstlane(unsigned v) { return simt::shfl_impl(v, simt::first_alive_lane()); }
static inline unsigned __builtin_amdgcn_ds_bpermute(int byteAddr, unsigned v) { return simt::shfl_impl(v, (byteAddr >> 2) & 63); }
static inline bool __builtin_amdgcn_inverse_ballot_w64(unsigned long long m) { return (m >> simt::cur->lane) & 1; }
#define ZHIP_SBFM64(width, offset) ((((width) & 63) ? (~0ull >> (64 - ((width) & 63))) : 0ull) << ((offset) & 63))    /* s_bfm_b64 */
#define ZHIP_WRITELANE(v, l, old) (simt::cur->lane == (int)(l) ? (unsigned)(v) : (unsigned)(old))    /* v_writelane_b32 */
static inline void __builtin_amdgcn_wave_barrier() { simt::wave_rendezvous(simt::K_WBAR, nullptr); }
static inline void __builtin_amdgcn_s_barrier();
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned m, unsigned v) { int l = simt::cur->lane; return v + __builtin_popcount(m & (l >= 32 ? 0xFFFFFFFFu : ((1u << l) - 1))); }
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned m, unsigned v) { int l = simt::cur->lane; return v + (l > 32 ? __builtin_popcount(m & ((1u << (l - 32)) - 1)) : 0); }
static inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)((((uint64_t)hi << 32) | lo) >> (8 * (sh & 3))); }
static inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)((((uint64_t)hi << 32) | lo) >> (sh & 31)); }
static inline unsigned __builtin_amdgcn_ubfe(unsigned v, unsigned off, unsigned w) { return w == 0 ? 0 : (v >> (off & 31)) & ((w >= 32) ? ~0u : ((1u << w) - 1)); }
static inline unsigned __builtin_amdgcn_sad_u8(unsigned a, unsigned b, unsigned c) { for (int i = 0; i < 4; i++) { int x = (a >> (8*i)) & 255, y = (b >> (8*i)) & 255; c += x > y ? x - y : y - x; } return c; }

static inline void __syncthreads()
{
    simt::Lane* me = simt::cur; simt::Group* g = me->g;
    unsigned const g0 = g->gen;
    if (++g->arrived == g->alive) { g->arrived = 0; g->gen++; }
    else while (g->gen == g0) simt::yield();
}
static inline void __builtin_amdgcn_s_barrier() { __syncthreads(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
#define __builtin_amdgcn_fence(order, scope) ((void)0)

static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
static inline unsigned __brev(unsigned v) { unsigned r = 0; for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i); return r; }

// fibers of one group run on one OS thread and never preempt each other inside an atomic; different groups may run
// on different OS threads, so global-memory atomics use real atomics.
template <typename T> static inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicSub(T* p, T v) { return __atomic_fetch_sub(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicAnd(T* p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicXor(T* p, T v) { return __atomic_fetch_xor(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; while (o < v && !__atomic_compare_exchange_n(p, &o, v, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = *p; while (o > v && !__atomic_compare_exchange_n(p, &o, v, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
template <typename T> static inline T atomicCAS(T* p, T c, T v) { __atomic_compare_exchange_n(p, &c, v, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return c; }

struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
