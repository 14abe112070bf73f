// zhip_parse_ext.h — gfx950 match finder for sources compressed with a dictionary in the reference's COPY mode
// (ZSTD_resetCCtx_byCopyingCDict, lib/compress/zstd_compress.c:2395-2470): sources above the attach cut-off (8 KB for a
// strategy-fast CDict, 16 KB for dfast, :2289-2315).  The working tables start as copies of the CDict's tables with the tags
// removed (:2379-2393), the dictionary content is the window's extDict segment (indices 2 .. P-1, P = dictLen + 2), the source
// is the prefix (indices P ..), and the block compressors are
//     ZSTD_compressBlock_fast_extDict_generic        lib/compress/zstd_fast.c:709-960
//     ZSTD_compressBlock_doubleFast_extDict_generic  lib/compress/zstd_double_fast.c:551-759
//
// HOW.  These sources are few and large compared with the attach-mode records, and their tables (the CDict's: 2^hashLog +
// 2^chainLog words, 384 KB at level 3) cannot live in LDS, so the work is spread the other way round: ONE LANE PER SOURCE, 64
// sources per wavefront, each lane walking the reference's loop for its own source with its own table pair in HBM
// (k_ext_init copies the CDict's tables, tags stripped, coalesced).  Every access is a dependent global round trip, so a
// source runs at CPU-like latency, but 64 x (waves in flight) of them run at once; a batch of thousands of 16-128 KB sources
// keeps the memory system busy, a single source does not (and is better served by the attach path's cut-off anyway).
// Output = the same ZhipSeq records / literal buffer / ZhipParse as the other match finders, so k_entropy takes over unchanged.
#pragma once
#include <hip/hip_runtime.h>
#include "zhip_common.h"
#include "zhip_parse.h"
#include "zhip_parse_dict.h"

namespace zhip {

// words of table memory one source needs in copy mode (the CDict's geometry)
__host__ __device__ inline size_t ext_table_words(uint32_t hashLog, uint32_t chainLog, uint32_t strategy)
{
    return ((size_t)1 << hashLog) + (strategy == ZHIP_STRAT_DFAST ? ((size_t)1 << chainLog) : 0);
}

__device__ __forceinline__ uint32_t ext_hash(const uint8_t* p, uint32_t hBits, uint32_t mls)   // zstd_compress_internal.h:820-862
{
    switch (mls) {
    default:
    case 4: return (ld32(p) * 2654435761U) >> (32 - hBits);
    case 5: return (uint32_t)(((ld64(p) << 24) * 889523592379ULL) >> (64 - hBits));
    case 6: return (uint32_t)(((ld64(p) << 16) * 227718039650203ULL) >> (64 - hBits));
    case 7: return (uint32_t)(((ld64(p) << 8) * 58295818150454627ULL) >> (64 - hBits));
    case 8: return (uint32_t)((ld64(p) * 0xCF1BBCDCB7A56463ULL) >> (64 - hBits));
    }
}
// ZSTD_count (zstd_compress_internal.h:771) by one lane: 8 bytes at a time while both sides have them
__device__ __forceinline__ uint32_t ext_count(const uint8_t* ip, const uint8_t* match, const uint8_t* iEnd)
{
    const uint8_t* const i0 = ip;
    while (ip + 8 <= iEnd) {
        uint64_t const x = ld64(ip) ^ ld64(match);
        if (x) return (uint32_t)(ip - i0) + ((uint32_t)(__ffsll((long long)x) - 1) >> 3);
        ip += 8; match += 8;
    }
    while (ip < iEnd && *ip == *match) { ip++; match++; }
    return (uint32_t)(ip - i0);
}
// ZSTD_count_2segments (:797): the match may run from the dictionary's end on into the source's first bytes
__device__ __forceinline__ uint32_t ext_count_2seg(const uint8_t* ip, const uint8_t* match, const uint8_t* iEnd, const uint8_t* mEnd, const uint8_t* iStart)
{
    const uint8_t* const vEnd = (ip + (mEnd - match) < iEnd) ? ip + (mEnd - match) : iEnd;
    uint32_t const k = ext_count(ip, match, vEnd);
    if (match + k != mEnd) return k;
    return k + ext_count(ip + k, iStart, iEnd);
}

// per-lane output (the job of ZSTD_storeSeq, zstd_compress_internal.h:671-728)
struct ExtOut { ZhipSeq* seqs; uint8_t* lits; uint32_t nbSeq, cap, longPos, longType, litPos; };
__device__ __forceinline__ void ext_store(ExtOut& o, const uint8_t* lit, uint32_t litLength, uint32_t offBase, uint32_t ml)
{
    uint32_t i = 0;
    for (; i + 8 <= litLength; i += 8) st64(o.lits + o.litPos + i, ld64(lit + i));
    for (; i < litLength; i++) o.lits[o.litPos + i] = lit[i];
    o.litPos += litLength;
    if (o.nbSeq >= o.cap) return;                              // cannot happen: matches are >= 4 bytes (zstd_compress.c:1690)
    uint32_t const mlBase = ml - 3;
    if (litLength > 0xFFFF) { o.longType = 1; o.longPos = o.nbSeq; }
    if (mlBase > 0xFFFF) { o.longType = 2; o.longPos = o.nbSeq; }
    ZhipSeq s; s.offBase = offBase; s.litLength = (uint16_t)litLength; s.mlBase = (uint16_t)mlBase;
    o.seqs[o.nbSeq++] = s;
}

struct ExtCtx {                 // one source, one lane
    const uint8_t* src; uint32_t n;
    const uint8_t* dict; uint32_t dictLen;      // content
    uint32_t* tabL; uint32_t* tabS;
    uint32_t hashLog, chainLog, mls, targetLength;
};
#define EXT_PTR(k) ((k) < P ? dictBase + (k) : base + (k))

// ZSTD_compressBlock_fast_extDict_generic (zstd_fast.c:709-960). returns the number of trailing literals
__device__ inline uint32_t ext_fast_source(const ExtCtx& c, ExtOut& out, uint32_t rep[3])
{
    uint32_t const hlog = c.hashLog, mls = c.mls;
    uint32_t* const T = c.tabL;
    uint32_t const stepSize = c.targetLength + !c.targetLength + 1;
    uint32_t const P = c.dictLen + 2, dictStartIndex = 2;
    const uint8_t* const src = c.src;
    const uint8_t* const base = src - P; const uint8_t* const dictBase = c.dict - 2;
    const uint8_t* const dictStart = c.dict; const uint8_t* const dictEnd = c.dict + c.dictLen;
    const uint8_t* const istart = src; const uint8_t* const iend = src + c.n; const uint8_t* const ilimit = iend - 8; const uint8_t* const prefixStart = src;
    const uint8_t* anchor = istart; const uint8_t* ip0 = istart; const uint8_t* ip1; const uint8_t* ip2; const uint8_t* ip3; const uint8_t* nextStep;
    const uint8_t* match0 = nullptr; const uint8_t* matchEnd = nullptr;
    uint32_t offset_1 = rep[0], offset_2 = rep[1], offsetSaved1 = 0, offsetSaved2 = 0, current0 = 0, idx, offcode = 0, hash0, hash1, step, mLength = 0;
    {   uint32_t const maxRep = (uint32_t)(ip0 - base) - dictStartIndex;                   // :764-768
        if (offset_2 >= maxRep) { offsetSaved2 = offset_2; offset_2 = 0; }
        if (offset_1 >= maxRep) { offsetSaved1 = offset_1; offset_1 = 0; }
    }
    for (;;) {                                                                               // _start
        int found = 0;
        step = stepSize; nextStep = ip0 + 128;
        ip1 = ip0 + 1; ip2 = ip0 + step; ip3 = ip2 + 1;
        if (ip3 >= ilimit) break;
        hash0 = ext_hash(ip0, hlog, mls); hash1 = ext_hash(ip1, hlog, mls);
        idx = T[hash0];
        do {
            {   uint32_t const current2 = (uint32_t)(ip2 - base), repIndex = current2 - offset_1;           // :790-817
                uint32_t rval;
                if (((uint32_t)(P - repIndex) >= 4) & (offset_1 > 0)) rval = ld32(EXT_PTR(repIndex)); else rval = ld32(ip2) ^ 1;
                current0 = (uint32_t)(ip0 - base); T[hash0] = current0;
                if (ld32(ip2) == rval) {
                    ip0 = ip2; match0 = EXT_PTR(repIndex); matchEnd = repIndex < P ? dictEnd : iend;
                    mLength = ip0[-1] == match0[-1];
                    ip0 -= mLength; match0 -= mLength;
                    offcode = 1; mLength += 4;
                    found = 2; break;
                }
            }
            if (idx >= dictStartIndex && ld32(EXT_PTR(idx)) == ld32(ip0)) { found = 1; break; }             // :819-829
            idx = T[hash1];                                                                                   // :831-846
            hash0 = hash1; hash1 = ext_hash(ip2, hlog, mls);
            ip0 = ip1; ip1 = ip2; ip2 = ip3;
            current0 = (uint32_t)(ip0 - base); T[hash0] = current0;
            if (idx >= dictStartIndex && ld32(EXT_PTR(idx)) == ld32(ip0)) { found = 1; break; }             // :848-858
            idx = T[hash1];                                                                                   // :860-880
            hash0 = hash1; hash1 = ext_hash(ip2, hlog, mls);
            ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step;
            if (ip2 >= nextStep) { step++; nextStep += 128; }
        } while (ip3 < ilimit);
        if (!found) break;                                                                                    // _cleanup
        if (found == 1) {                                                                                     // _offset :899-915
            uint32_t const offset = current0 - idx;
            const uint8_t* const low = idx < P ? dictStart : prefixStart;
            matchEnd = idx < P ? dictEnd : iend;
            match0 = EXT_PTR(idx);
            offset_2 = offset_1; offset_1 = offset;
            offcode = offset + 3; mLength = 4;
            while (((ip0 > anchor) & (match0 > low)) && ip0[-1] == match0[-1]) { ip0--; match0--; mLength++; }
        }
        mLength += ext_count_2seg(ip0 + mLength, match0 + mLength, iend, matchEnd, prefixStart);            // _match :917-957
        ext_store(out, anchor, (uint32_t)(ip0 - anchor), offcode, mLength);
        ip0 += mLength; anchor = ip0;
        if (ip1 < ip0) T[hash1] = (uint32_t)(ip1 - base);
        if (ip0 <= ilimit) {
            T[ext_hash(base + current0 + 2, hlog, mls)] = current0 + 2;
            T[ext_hash(ip0 - 2, hlog, mls)] = (uint32_t)(ip0 - 2 - base);
            while (ip0 <= ilimit) {
                uint32_t const repIndex2 = (uint32_t)(ip0 - base) - offset_2;
                const uint8_t* const repMatch2 = EXT_PTR(repIndex2);
                if ((((uint32_t)((P - 1) - repIndex2) >= 3) & (offset_2 > 0)) && ld32(repMatch2) == ld32(ip0)) {
                    const uint8_t* const repEnd2 = repIndex2 < P ? dictEnd : iend;
                    uint32_t const rl = ext_count_2seg(ip0 + 4, repMatch2 + 4, iend, repEnd2, prefixStart) + 4;
                    uint32_t const t = offset_2; offset_2 = offset_1; offset_1 = t;
                    ext_store(out, anchor, 0, 1, rl);
                    T[ext_hash(ip0, hlog, mls)] = (uint32_t)(ip0 - base);
                    ip0 += rl; anchor = ip0;
                    continue;
                }
                break;
            }
        }
    }
    offsetSaved2 = (offsetSaved1 != 0 && offset_1 != 0) ? offsetSaved1 : offsetSaved2;
    rep[0] = offset_1 ? offset_1 : offsetSaved1;
    rep[1] = offset_2 ? offset_2 : offsetSaved2;
    return (uint32_t)(iend - anchor);
}

// ZSTD_compressBlock_doubleFast_extDict_generic (zstd_double_fast.c:551-759)
__device__ inline uint32_t ext_dfast_source(const ExtCtx& c, ExtOut& out, uint32_t rep[3])
{
    uint32_t const hBitsL = c.hashLog, hBitsS = c.chainLog, mls = c.mls;
    uint32_t* const hashLong = c.tabL; uint32_t* const hashSmall = c.tabS;
    uint32_t const P = c.dictLen + 2, dictStartIndex = 2;
    const uint8_t* const src = c.src;
    const uint8_t* const base = src - P; const uint8_t* const dictBase = c.dict - 2;
    const uint8_t* const dictStart = c.dict; const uint8_t* const dictEnd = c.dict + c.dictLen;
    const uint8_t* const istart = src; const uint8_t* const iend = src + c.n; const uint8_t* const ilimit = iend - 8; const uint8_t* const prefixStart = src;
    const uint8_t* ip = istart; const uint8_t* anchor = istart;
    uint32_t offset_1 = rep[0], offset_2 = rep[1];
    while (ip < ilimit) {
        uint32_t const hSmall = ext_hash(ip, hBitsS, mls), hLong = ext_hash(ip, hBitsL, 8);
        uint32_t const matchIndex = hashSmall[hSmall], matchLongIndex = hashLong[hLong];
        const uint8_t* match = EXT_PTR(matchIndex); const uint8_t* matchLong = EXT_PTR(matchLongIndex);
        uint32_t const curr = (uint32_t)(ip - base), repIndex = curr + 1 - offset_1;
        const uint8_t* const repMatch = EXT_PTR(repIndex);
        uint32_t mLength;
        hashSmall[hSmall] = curr; hashLong[hLong] = curr;
        if (((uint32_t)((P - 1) - repIndex) >= 3) && (offset_1 <= curr + 1 - dictStartIndex) && ld32(repMatch) == ld32(ip + 1)) {   // :613
            const uint8_t* const repEnd = repIndex < P ? dictEnd : iend;
            mLength = ext_count_2seg(ip + 1 + 4, repMatch + 4, iend, repEnd, prefixStart) + 4;
            ip++;
            ext_store(out, anchor, (uint32_t)(ip - anchor), 1, mLength);
        } else {
            if (matchLongIndex > dictStartIndex && ld64(matchLong) == ld64(ip)) {                 // :621
                const uint8_t* const matchEnd = matchLongIndex < P ? dictEnd : iend;
                const uint8_t* const low = matchLongIndex < P ? dictStart : prefixStart;
                mLength = ext_count_2seg(ip + 8, matchLong + 8, iend, matchEnd, prefixStart) + 8;
                uint32_t const offset = curr - matchLongIndex;
                while (ip > anchor && matchLong > low && ip[-1] == matchLong[-1]) { ip--; matchLong--; mLength++; }
                offset_2 = offset_1; offset_1 = offset;
                ext_store(out, anchor, (uint32_t)(ip - anchor), offset + 3, mLength);
            } else if (matchIndex > dictStartIndex && ld32(match) == ld32(ip)) {                  // :633
                uint32_t const h3 = ext_hash(ip + 1, hBitsL, 8), matchIndex3 = hashLong[h3];
                const uint8_t* match3 = EXT_PTR(matchIndex3);
                uint32_t offset;
                hashLong[h3] = curr + 1;
                if (matchIndex3 > dictStartIndex && ld64(match3) == ld64(ip + 1)) {
                    const uint8_t* const matchEnd = matchIndex3 < P ? dictEnd : iend;
                    const uint8_t* const low = matchIndex3 < P ? dictStart : prefixStart;
                    mLength = ext_count_2seg(ip + 9, match3 + 8, iend, matchEnd, prefixStart) + 8;
                    ip++;
                    offset = curr + 1 - matchIndex3;
                    while (ip > anchor && match3 > low && ip[-1] == match3[-1]) { ip--; match3--; mLength++; }
                } else {
                    const uint8_t* const matchEnd = matchIndex < P ? dictEnd : iend;
                    const uint8_t* const low = matchIndex < P ? dictStart : prefixStart;
                    mLength = ext_count_2seg(ip + 4, match + 4, iend, matchEnd, prefixStart) + 4;
                    offset = curr - matchIndex;
                    while (ip > anchor && match > low && ip[-1] == match[-1]) { ip--; match--; mLength++; }
                }
                offset_2 = offset_1; offset_1 = offset;
                ext_store(out, anchor, (uint32_t)(ip - anchor), offset + 3, mLength);
            } else { ip += ((ip - anchor) >> 8) + 1; continue; }
        }
        ip += mLength; anchor = ip;
        if (ip <= ilimit) {                                                                       // :677
            uint32_t const ins = curr + 2;
            hashLong[ext_hash(base + ins, hBitsL, 8)] = ins;
            hashLong[ext_hash(ip - 2, hBitsL, 8)] = (uint32_t)(ip - 2 - base);
            hashSmall[ext_hash(base + ins, hBitsS, mls)] = ins;
            hashSmall[ext_hash(ip - 1, hBitsS, mls)] = (uint32_t)(ip - 1 - base);
            while (ip <= ilimit) {
                uint32_t const current2 = (uint32_t)(ip - base), repIndex2 = current2 - offset_2;
                const uint8_t* const repMatch2 = EXT_PTR(repIndex2);
                if (((uint32_t)((P - 1) - repIndex2) >= 3) && (offset_2 <= current2 - dictStartIndex) && ld32(repMatch2) == ld32(ip)) {
                    const uint8_t* const repEnd2 = repIndex2 < P ? dictEnd : iend;
                    uint32_t const rl = ext_count_2seg(ip + 4, repMatch2 + 4, iend, repEnd2, prefixStart) + 4;
                    uint32_t const t = offset_2; offset_2 = offset_1; offset_1 = t;
                    ext_store(out, anchor, 0, 1, rl);
                    hashSmall[ext_hash(ip, hBitsS, mls)] = current2;
                    hashLong[ext_hash(ip, hBitsL, 8)] = current2;
                    ip += rl; anchor = ip;
                    continue;
                }
                break;
            }
        }
    }
    rep[0] = offset_1; rep[1] = offset_2;
    return (uint32_t)(iend - anchor);
}
#undef EXT_PTR

// one lane = one source of the copy-mode list (ext[] = indices into units[])
__device__ inline void parse_ext_source(const uint8_t* src, const ZhipUnit& u, const ZhipCDictDev& cd, uint32_t* tabs,
                                        ZhipSeq* seqs, uint32_t seqCap, uint8_t* lits, ZhipParse* meta)
{
    ExtCtx c; c.src = src; c.n = u.srcLen; c.dict = cd.content; c.dictLen = cd.len;
    c.tabL = tabs; c.tabS = tabs + ((size_t)1 << u.hashLog);
    c.hashLog = u.hashLog; c.chainLog = u.chainLog; c.mls = u.minMatch; c.targetLength = u.targetLength;
    ExtOut out; out.seqs = seqs; out.lits = lits; out.nbSeq = 0; out.cap = seqCap; out.longPos = 0; out.longType = 0; out.litPos = 0;
    uint32_t rep[3] = { cd.rep[0], cd.rep[1], cd.rep[2] };
    uint32_t last;
    if (u.srcLen < 8) last = u.srcLen;
    else last = u.strategy == ZHIP_STRAT_FAST ? ext_fast_source(c, out, rep) : ext_dfast_source(c, out, rep);
    {   const uint8_t* const lit = src + u.srcLen - last;                                        // trailing literals (zstd_compress.c:3365)
        for (uint32_t i = 0; i < last; i++) lits[out.litPos + i] = lit[i];
        out.litPos += last;
    }
    meta->nbSeq = out.nbSeq; meta->lastLits = last; meta->longPos = out.longPos; meta->longType = out.longType;
    meta->rep[0] = rep[0]; meta->rep[1] = rep[1]; meta->rep[2] = rep[2];
    meta->status = 0; meta->litSize = out.litPos; meta->pad0 = 0;
}

}  // namespace zhip
