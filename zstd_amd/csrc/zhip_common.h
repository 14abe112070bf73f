// zhip_common.h — data layout shared by the gfx950 kernels and the host side of libzstd_hip.
//
// Domain vocabulary follows the reference (facebook/zstd): a *unit* is one independently compressed chunk of
// source (<= 128 KB, the reference's ZSTD_BLOCKSIZE_MAX) that becomes one frame holding one block; a *sequence*
// is (litLength, matchLength, offBase) exactly as lib/common/zstd_internal.h:281-311 (seqDef) stores it.
#pragma once
#include <stdint.h>

#define ZHIP_UNIT_MAX        131072u          /* lib/zstd.h ZSTD_BLOCKSIZE_MAX */
#define ZHIP_SEQ_CAP         (ZHIP_UNIT_MAX / 4 + 8)   /* fast/dfast matches are >= 4 bytes (zstd_compress.c:1690) */
#define ZHIP_OUT_STRIDE      (ZHIP_UNIT_MAX + 512 + 32) /* >= ZSTD_compressBound(128 KB) + frame header */
#define ZHIP_LIT_STRIDE      (ZHIP_UNIT_MAX + 64)
#define ZHIP_UNIT_COPYMODE    0xC0u

enum { ZHIP_STRAT_FAST = 1, ZHIP_STRAT_DFAST = 2, ZHIP_STRAT_GREEDY = 3, ZHIP_STRAT_LAZY = 4, ZHIP_STRAT_LAZY2 = 5 };   /* lib/zstd.h:328-337 */

// one entry per unit, filled by the host, read by every kernel
struct ZhipUnit {
    uint64_t srcOff;        // byte offset of the unit in the source buffer
    uint32_t srcLen;        // <= ZHIP_UNIT_MAX
    uint8_t  windowLog, chainLog, hashLog, minMatch;
    uint8_t  strategy, searchLog, litMode /* 1: literals stay raw (negative levels) */, pad0 /* records path: ZHIP_UNIT_COPYMODE = the dictionary is copied, not attached */;
    uint32_t targetLength;
    uint32_t rowLog;        // greedy / lazy / lazy2: 0 = hash-chain matcher, 4 / 5 / 6 = row-hash matcher with rows of 2^rowLog entries
    uint32_t pad1;
};

// where a unit's intermediate results live: offsets into the context's arenas, filled by the host.  Full-size units use the
// fixed strides above; small records (dictionary path) are packed back to back.
struct ZhipSlot {
    uint64_t seqOff;        // ZhipSeq index into the sequence arena; the unit's three code arrays start at 3*seqOff in the
                            // code arena (uint16_t), each seqCap entries long
    uint64_t litOff;        // byte offset into the literal arena
    uint64_t outOff;        // byte offset into the output-slot arena (16-byte aligned)
    uint32_t seqCap;        // capacity in sequences
    uint32_t pad0;
};

// 8-byte sequence record (same packing as the reference's seqDef + its single long-length escape)
struct ZhipSeq {
    uint32_t offBase;       // 1..3 repcode id, else offset + 3
    uint16_t litLength;     // low 16 bits (see ZhipParse.longPos)
    uint16_t mlBase;        // matchLength - 3, low 16 bits
};

// written by the parse kernel, read by the entropy kernel
struct ZhipParse {
    uint32_t nbSeq;
    uint32_t lastLits;      // literals after the last match
    uint32_t longPos;       // index of the one sequence whose length overflowed 16 bits
    uint32_t longType;      // 0 none, 1 literal length, 2 match length (zstd_internal.h:296-300)
    uint32_t rep[3];        // repcode history after the unit
    uint32_t status;        // 0 ok
    uint32_t litSize;       // literals the match finder copied to the unit's literal buffer (incl. lastLits)
    uint32_t pad0;
};

// ------------------------------------------------------------------ wave-uniform control flow (kernels that use workgroup barriers)
// A value every lane of a wavefront holds alike but the compiler cannot prove so (anything loaded from LDS or memory) makes each branch on it
// an EXEC-masked one, and an EXEC-masked loop around s_barrier is at the mercy of how the compiler orders the "divergent" paths (the
// round-2 and round-3 GPU-only stalls).  ZHIP_UNIFORM moves such a value into a scalar register: the branch becomes
// s_cbranch_scc for the whole wavefront.  ZHIP_CONVERGE is a convergent no-op: code on either side of it is not merged or threaded across.
// tests/test_isa_checks.py holds the compiled kernels to it (no EXEC-conditional branch may span an s_barrier).
#define ZHIP_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(x)))
#define ZHIP_CONVERGE() __builtin_amdgcn_wave_barrier()

// ------------------------------------------------------------------ optional phase profiling (scripts/prof_phases.py)
// Compiled only into the measurement variant of the library (-DZHIP_PROF, zstd_amd/libzstd_hip_prof.so): thread 0 of
// each workgroup accumulates s_memtime deltas per phase and adds them to g_prof at the end.  The product build
// expands all of this to nothing.
#ifdef ZHIP_PROF
namespace zhip { __device__ unsigned long long g_prof[32]; __device__ unsigned long long g_wph[64]; }   /* g_wph: ZSTD_fast window phases, [0,32) ticks, [32,64) visits */
#define ZPROF_DECL uint64_t zp_last_ = __builtin_amdgcn_s_memtime(); uint64_t zp_acc_[16] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
#define ZPROF(i) do { uint64_t const t_ = __builtin_amdgcn_s_memtime(); zp_acc_[i] += t_ - zp_last_; zp_last_ = t_; } while (0)
#define ZPROF_COUNT(i, v) do { zp_acc_[i] += (uint64_t)(v); } while (0)
#define ZPROF_FLUSH(base) do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 16; i_++) atomicAdd(&zhip::g_prof[(base) + i_], (unsigned long long)zp_acc_[i_]); } while (0)
#define ZPROF_JOB_BEGIN uint64_t zj_ = __builtin_amdgcn_s_memtime();
#define ZPROF_JOB_MARK(slot) do { uint64_t const t_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) atomicAdd(&zhip::g_prof[slot], (unsigned long long)(t_ - zj_)); zj_ = t_; } while (0)
#else
#define ZPROF_JOB_BEGIN
#define ZPROF_JOB_MARK(slot) do { } while (0)
#define ZPROF_DECL
#define ZPROF(i) do { } while (0)
#define ZPROF_COUNT(i, v) do { } while (0)
#define ZPROF_FLUSH(base) do { } while (0)
#endif
