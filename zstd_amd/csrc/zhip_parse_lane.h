// zhip_parse_lane.h — gfx950 match finders for LARGE batches of units: ONE LANE PER UNIT (round 3).
//
// WHAT: exactly the sequences of ZSTD_compressBlock_fast_noDict_generic (lib/compress/zstd_fast.c:192-423) and
// ZSTD_compressBlock_doubleFast_noDict_generic (lib/compress/zstd_double_fast.c:105-323) for a unit with no history — the same
// contract as zhip_parse.h / zhip_parse_dfast.h, the same output (ZhipSeq records, literal buffer, ZhipParse).
//
// WHY a second form.  The wave-per-unit kernels spend a whole wavefront on one unit's serial chain of decisions: 2 304 (fast, LDS
// table) or 4 096 (dfast, registers) units are in flight, each paced by dependent memory round trips (dfast: 12.7 G line fetches per
// second against the ~54 G/s the memory system sustains; fast on dense-match data: ~3 000 wave-cycles per sequence of issue).  A batch
// of 100 000 units (BASELINE configs[2]: 13 GiB) has enough independent chains to fill the machine the other way round: every LANE walks
// its own unit with its own table(s) in HBM — 64 chains per wavefront, every memory instruction 64 independent requests — so the
// kernel is paced by the memory system's request rate, not by one chain's latency, and the instruction count per sequence drops by the
// SIMT width.  Measured first on the dictionary copy mode (zhip_parse_ext.h, the same shape): 24 GB/s with only 25 000 lanes.
// The price: a unit runs at CPU-like latency (~1 MB/s), so the form only pays when there are tens of thousands of units; the host picks
// it per call ($ZHIP_LANE_MIN_UNITS, zhip_lib.hip) and the wave-per-unit kernels keep the small batches (the 1 GiB headline).
//
// HOW.  Tables: 32-bit entries `position | tag << 17` in HBM, zeroed by one memset for all units (a lane zeroing its own table would
// write 64 scattered lines per instruction).  The tag is a function of exactly the bytes the reference compares at a candidate, so a
// mismatch proves the compare fails and the candidate's bytes — a random line — are not fetched.  Positions are unit-relative, 0 = empty
// (position 0 is never inserted: ip += (ip == prefixStart), zstd_fast.c:238 / zstd_double_fast.c:157).
#pragma once
#include <hip/hip_runtime.h>
#include "zhip_common.h"
#include "zhip_parse.h"
#include "zhip_parse_dfast.h"

namespace zhip {

#define ZHIP_UNIT_LANE 1u            /* ZhipUnit.pad1: this unit belongs to k_parse_lane (the wave-per-unit kernels skip it) */

__device__ __forceinline__ uint32_t lane_tag4(uint32_t first4) { return (first4 * 2654435761U) >> 17; }      // 15 bits of the 4 bytes a ZSTD_fast candidate is compared on
// words of table memory one unit needs in this form
__host__ __device__ inline size_t lane_table_words(uint32_t hashLog, uint32_t chainLog, uint32_t strategy)
{
    return ((size_t)1 << hashLog) + (strategy == ZHIP_STRAT_DFAST ? ((size_t)1 << chainLog) : 0);
}

// ZSTD_count (zstd_compress_internal.h:771) by one lane over unit-relative positions: equal bytes of src[a..) and src[b..), a > b
__device__ __forceinline__ uint32_t ln_count(const uint8_t* src, uint32_t a, uint32_t b, uint32_t n)
{
    uint32_t const a0 = a;
    while (a + 8 <= n) {
        uint64_t const x = ld64(src + a) ^ ld64(src + b);
        if (x) return a - a0 + ((uint32_t)(__ffsll((long long)x) - 1) >> 3);
        a += 8; b += 8;
    }
    while (a < n && src[a] == src[b]) { a++; b++; }
    return a - a0;
}

struct LnOut { ZhipSeq* seqs; uint8_t* lits; uint32_t nbSeq, cap, longPos, longType, litPos; };
// ZSTD_storeSeq (zstd_compress_internal.h:671-728): the literals [from, from+ll) and the sequence record
__device__ __forceinline__ void ln_store(LnOut& o, const uint8_t* src, uint32_t from, uint32_t litLength, uint32_t offBase, uint32_t ml)
{
    uint32_t i = 0;
    for (; i + 8 <= litLength; i += 8) st64(o.lits + o.litPos + i, ld64(src + from + i));
    for (; i < litLength; i++) o.lits[o.litPos + i] = src[from + i];
    o.litPos += litLength;
    if (o.nbSeq >= o.cap) return;                              // cannot happen: matches are >= 4 bytes (zstd_compress.c:1690)
    uint32_t const mlBase = ml - 3;
    if (litLength > 0xFFFF) { o.longType = 1; o.longPos = o.nbSeq; }
    if (mlBase > 0xFFFF) { o.longType = 2; o.longPos = o.nbSeq; }
    ZhipSeq s; s.offBase = offBase; s.litLength = (uint16_t)litLength; s.mlBase = (uint16_t)mlBase;
    o.seqs[o.nbSeq++] = s;
}

// ------------------------------------------------------------------ ZSTD_fast, zstd_fast.c:192-423
template <uint32_t MLS>
__device__ inline uint32_t lane_fast_unit(const uint8_t* __restrict__ src, uint32_t n, uint32_t hlog, uint32_t stepSize, uint32_t* __restrict__ T,
                                          LnOut& out, uint32_t rep[3])
{
    uint32_t const hshift = 32 - hlog;
    uint32_t anchor = 0, ip0 = 1;                                           // :238 ip0 += (ip0 == prefixStart)
    uint32_t rep1 = rep[0], rep2 = rep[1], saved1 = 0, saved2 = 0;
    {   uint32_t const maxRep = ip0;                                        // :240-244 (windowLow = 0)
        if (rep2 > maxRep) { saved2 = rep2; rep2 = 0; }
        if (rep1 > maxRep) { saved1 = rep1; rep1 = 0; }
    }
    if (n >= 12) {
    uint32_t const ilimit = n - 8;
#define LF_ENTRY(p, b4) ((p) | (lane_tag4(b4) << 17))
#define LF_HIT(e, b4, p) (((e) & ZHIP_DF_POS) != 0 && ((e) >> 17) == lane_tag4(b4) && ld32(src + ((e) & ZHIP_DF_POS)) == (b4))
    for (;;) {                                                              // _start
        uint32_t step = stepSize, nextStep = ip0 + 128;
        uint32_t ip1 = ip0 + 1, ip2 = ip0 + step, ip3 = ip2 + 1;
        if (ip3 >= ilimit) break;                                           // :257
        uint64_t b0 = ld64(src + ip0), b1 = ld64(src + ip1);
        uint32_t hash0 = hash_pos<MLS>(b0, hshift), hash1 = hash_pos<MLS>(b1, hshift);
        uint32_t e = T[hash0];
        uint32_t current0 = 0, offcode = 0, mLength = 0, match0 = 0;
        int found = 0;
        do {
            uint32_t const c2 = ld32(src + ip2);
            uint32_t const rval = rep1 ? ld32(src + ip2 - rep1) : c2 ^ 1u;  // :268
            current0 = ip0; T[hash0] = LF_ENTRY(ip0, (uint32_t)b0);         // :271-272
            if (c2 == rval) {                                               // :275-290 repcode at ip2
                ip0 = ip2; match0 = ip0 - rep1;
                mLength = src[ip0 - 1] == src[match0 - 1] ? 1u : 0u;
                ip0 -= mLength; match0 -= mLength;
                offcode = 1; mLength += 4;
                T[hash1] = LF_ENTRY(ip1, (uint32_t)b1);
                found = 2; break;
            }
            if (LF_HIT(e, (uint32_t)b0, ip0)) {                             // :292-299
                T[hash1] = LF_ENTRY(ip1, (uint32_t)b1);
                found = 1; break;
            }
            e = T[hash1];                                                   // :302
            {   uint64_t const b2 = ld64(src + ip2);
                hash0 = hash1; b0 = b1; hash1 = hash_pos<MLS>(b2, hshift); b1 = b2; }
            ip0 = ip1; ip1 = ip2; ip2 = ip3;
            current0 = ip0; T[hash0] = LF_ENTRY(ip0, (uint32_t)b0);         // :313-314
            if (LF_HIT(e, (uint32_t)b0, ip0)) {                             // :317-326
                if (step <= 4) T[hash1] = LF_ENTRY(ip1, (uint32_t)b1);
                found = 1; break;
            }
            e = T[hash1];                                                   // :329
            {   uint64_t const b2 = ld64(src + ip2);
                hash0 = hash1; b0 = b1; hash1 = hash_pos<MLS>(b2, hshift); b1 = b2; }
            ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step;       // :336-339
            if (ip2 >= nextStep) { step++; nextStep += 128; }               // :342-346
        } while (ip3 < ilimit);
        if (!found) break;                                                  // _cleanup
        if (found == 1) {                                                   // _offset :377-391
            match0 = e & ZHIP_DF_POS;
            rep2 = rep1; rep1 = ip0 - match0;
            offcode = rep1 + 3; mLength = 4;
            while (ip0 > anchor && match0 > 0 && src[ip0 - 1] == src[match0 - 1]) { ip0--; match0--; mLength++; }
        }
        mLength += ln_count(src, ip0 + mLength, match0 + mLength, n);       // _match :396
        ln_store(out, src, anchor, ip0 - anchor, offcode, mLength);
        ip0 += mLength; anchor = ip0;
        if (ip0 <= ilimit) {                                                // :403-420
            {   uint64_t const ba = ld64(src + current0 + 2), bb = ld64(src + ip0 - 2);
                T[hash_pos<MLS>(ba, hshift)] = LF_ENTRY(current0 + 2, (uint32_t)ba);
                T[hash_pos<MLS>(bb, hshift)] = LF_ENTRY(ip0 - 2, (uint32_t)bb); }
            if (rep2 > 0) {
                while (ip0 <= ilimit) {
                    uint64_t const bi = ld64(src + ip0);
                    if ((uint32_t)bi != ld32(src + ip0 - rep2)) break;
                    uint32_t const rLength = ln_count(src, ip0 + 4, ip0 + 4 - rep2, n) + 4;
                    {   uint32_t const t = rep2; rep2 = rep1; rep1 = t; }
                    T[hash_pos<MLS>(bi, hshift)] = LF_ENTRY(ip0, (uint32_t)bi);
                    ln_store(out, src, anchor, 0, 1, rLength);
                    ip0 += rLength; anchor = ip0;
                }
            }
        }
    }
#undef LF_ENTRY
#undef LF_HIT
    }
    saved2 = (saved1 != 0 && rep1 != 0) ? saved1 : saved2;                  // :368-375
    rep[0] = rep1 ? rep1 : saved1; rep[1] = rep2 ? rep2 : saved2;
    return n - anchor;
}

// ------------------------------------------------------------------ ZSTD_dfast, zstd_double_fast.c:105-323
template <uint32_t MLS>
__device__ inline uint32_t lane_dfast_unit(const uint8_t* __restrict__ src, uint32_t n, uint32_t hBitsL, uint32_t hBitsS,
                                           uint32_t* __restrict__ L, uint32_t* __restrict__ S, LnOut& out, uint32_t rep[3])
{
    uint32_t const shL = 32 - hBitsL, shS = 32 - hBitsS;
    uint32_t anchor = 0, ip = 1;                                            // :157
    uint32_t off1 = rep[0], off2 = rep[1], saved1 = 0, saved2 = 0;
    {   uint32_t const maxRep = ip;                                         // :158-164
        if (off2 > maxRep) { saved2 = off2; off2 = 0; }
        if (off1 > maxRep) { saved1 = off1; off1 = 0; }
    }
    if (n >= 9) {
    uint32_t const ilimit = n - 8;
#define LD_E(p, tg) ((p) | ((tg) << 17))
    for (;;) {                                                              // one turn per match (:167)
        uint32_t step = 1, nextStep = ip + 256, ip1 = ip + 1;
        if (ip1 > ilimit) break;                                            // :172
        uint64_t b0 = ld64(src + ip);
        uint32_t vL0 = mulhi64_top32(b0, 0xCF1BBCDCB7A56463ULL);
        uint32_t eL0 = L[vL0 >> shL];
        uint32_t curr = 0, mLength = 0, offset = 0, hl1 = 0, mstart = 0, tgL1 = 0;
        int kind = 0;                                                       // 1 repcode, 2 match
        do {
            uint32_t const hs0 = hash_pos<MLS>(b0, shS);
            uint32_t const eS0 = S[hs0];
            uint64_t const b1 = ld64(src + ip1);
            uint32_t const vL1 = mulhi64_top32(b1, 0xCF1BBCDCB7A56463ULL);
            hl1 = vL1 >> shL; tgL1 = df_tag_long(vL1);
            curr = ip;
            L[vL0 >> shL] = LD_E(curr, df_tag_long(vL0)); S[hs0] = LD_E(curr, df_tag_short((uint32_t)b0));      // :187
            uint32_t const eL1 = L[hl1];                                    // :213, read right behind the update (ip and ip+1 may share a long hash) so that it overlaps the checks below
            if (off1 > 0 && ld32(src + ip + 1 - off1) == (uint32_t)(b0 >> 8)) {                               // :190 repcode at ip+1
                mLength = ln_count(src, ip + 5, ip + 5 - off1, n) + 4;
                mstart = ip + 1;
                kind = 1; break;
            }
            {   uint32_t const m = eL0 & ZHIP_DF_POS;                       // :200-209 long match at ip
                if (m != 0 && (eL0 >> 17) == df_tag_long(vL0) && ld64(src + m) == b0) {
                    mLength = ln_count(src, ip + 8, m + 8, n) + 8;
                    offset = ip - m; mstart = ip;
                    uint32_t mm = m;
                    while (mstart > anchor && mm > 0 && src[mstart - 1] == src[mm - 1]) { mstart--; mm--; mLength++; }
                    kind = 2; break;
                }
            }
            {   uint32_t const m = eS0 & ZHIP_DF_POS;                       // :214-220 short match at ip -> _search_next_long
                if (m != 0 && (eS0 >> 17) == df_tag_short((uint32_t)b0) && ld32(src + m) == (uint32_t)b0) {
                    mLength = ln_count(src, ip + 4, m + 4, n) + 4;
                    offset = ip - m; mstart = ip;
                    uint32_t mm = m;
                    uint32_t const m1 = eL1 & ZHIP_DF_POS;                  // :251-264 long match at ip+1 (index > lowest)
                    if (m1 != 0 && (eL1 >> 17) == tgL1 && ld64(src + m1) == b1) {
                        uint32_t const l1len = ln_count(src, ip1 + 8, m1 + 8, n) + 8;
                        if (l1len > mLength) { mstart = ip1; mLength = l1len; offset = ip1 - m1; mm = m1; }
                    }
                    while (mstart > anchor && mm > 0 && src[mstart - 1] == src[mm - 1]) { mstart--; mm--; mLength++; }     // :267
                    kind = 2; break;
                }
            }
            if (ip1 >= nextStep) { step++; nextStep += 256; }               // :222-227
            ip = ip1; ip1 += step;
            b0 = b1; vL0 = vL1; eL0 = eL1;
        } while (ip1 <= ilimit);
        if (!kind) break;                                                   // _cleanup
        if (kind == 2) {                                                    // _match_found :270-291
            off2 = off1; off1 = offset;
            if (step < 4) L[hl1] = LD_E(ip1, tgL1);                         // (ip1 is still the position behind curr)
            ln_store(out, src, anchor, mstart - anchor, offset + 3, mLength);
        } else ln_store(out, src, anchor, mstart - anchor, 1, mLength);
        ip = mstart + mLength; anchor = ip;                                 // _match_stored :293-297
        if (ip <= ilimit) {
            {   uint64_t const ba = ld64(src + curr + 2), bb = ld64(src + ip - 2), bc = ld64(src + ip - 1);      // :300-310
                uint32_t const va = mulhi64_top32(ba, 0xCF1BBCDCB7A56463ULL), vb = mulhi64_top32(bb, 0xCF1BBCDCB7A56463ULL);
                L[va >> shL] = LD_E(curr + 2, df_tag_long(va));
                L[vb >> shL] = LD_E(ip - 2, df_tag_long(vb));
                S[hash_pos<MLS>(ba, shS)] = LD_E(curr + 2, df_tag_short((uint32_t)ba));
                S[hash_pos<MLS>(bc, shS)] = LD_E(ip - 1, df_tag_short((uint32_t)bc)); }
            while (ip <= ilimit && off2 > 0) {                              // :313-327
                uint64_t const bi = ld64(src + ip);
                if ((uint32_t)bi != ld32(src + ip - off2)) break;
                uint32_t const rLength = ln_count(src, ip + 4, ip + 4 - off2, n) + 4;
                {   uint32_t const t = off2; off2 = off1; off1 = t; }
                uint32_t const vi = mulhi64_top32(bi, 0xCF1BBCDCB7A56463ULL);
                S[hash_pos<MLS>(bi, shS)] = LD_E(ip, df_tag_short((uint32_t)bi));
                L[vi >> shL] = LD_E(ip, df_tag_long(vi));
                ln_store(out, src, anchor, 0, 1, rLength);
                ip += rLength; anchor = ip;
            }
        }
    }
#undef LD_E
    }
    saved2 = (saved1 != 0 && off1 != 0) ? saved1 : saved2;                  // :248-256
    rep[0] = off1 ? off1 : saved1; rep[1] = off2 ? off2 : saved2;
    return n - anchor;
}

// one lane = one unit.  tabs: the unit's table(s), zeroed (long table first for dfast)
__device__ inline void parse_lane_unit(const uint8_t* __restrict__ src, const ZhipUnit& u, uint32_t* __restrict__ tabs,
                                       ZhipSeq* seqs, uint32_t seqCap, uint8_t* lits, ZhipParse* meta)
{
    LnOut out; out.seqs = seqs; out.lits = lits; out.nbSeq = 0; out.cap = seqCap; out.longPos = 0; out.longType = 0; out.litPos = 0;
    uint32_t rep[3] = { 1, 4, 8 };
    uint32_t const n = u.srcLen;
    uint32_t last;
    if (u.strategy == ZHIP_STRAT_FAST) {
        uint32_t const stepSize = u.targetLength + !u.targetLength + 1;
        switch (u.minMatch) {
        case 5:  last = lane_fast_unit<5>(src, n, u.hashLog, stepSize, tabs, out, rep); break;
        case 6:  last = lane_fast_unit<6>(src, n, u.hashLog, stepSize, tabs, out, rep); break;
        case 7:  last = lane_fast_unit<7>(src, n, u.hashLog, stepSize, tabs, out, rep); break;
        case 8:  last = lane_fast_unit<8>(src, n, u.hashLog, stepSize, tabs, out, rep); break;
        default: last = lane_fast_unit<4>(src, n, u.hashLog, stepSize, tabs, out, rep); break;
        }
    } else {
        uint32_t* const S = tabs + ((size_t)1 << u.hashLog);
        switch (u.minMatch) {
        case 5:  last = lane_dfast_unit<5>(src, n, u.hashLog, u.chainLog, tabs, S, out, rep); break;
        case 6:  last = lane_dfast_unit<6>(src, n, u.hashLog, u.chainLog, tabs, S, out, rep); break;
        case 7:  last = lane_dfast_unit<7>(src, n, u.hashLog, u.chainLog, tabs, S, out, rep); break;
        case 8:  last = lane_dfast_unit<8>(src, n, u.hashLog, u.chainLog, tabs, S, out, rep); break;
        default: last = lane_dfast_unit<4>(src, n, u.hashLog, u.chainLog, tabs, S, out, rep); break;
        }
    }
    {   uint32_t const from = n - last;                                     // trailing literals (zstd_compress.c:3365)
        uint32_t i = 0;
        for (; i + 8 <= last; i += 8) st64(lits + out.litPos + i, ld64(src + from + i));
        for (; i < last; i++) lits[out.litPos + i] = src[from + i];
        out.litPos += last;
    }
    meta->nbSeq = out.nbSeq; meta->lastLits = last; meta->longPos = out.longPos; meta->longType = out.longType;
    meta->rep[0] = rep[0]; meta->rep[1] = rep[1]; meta->rep[2] = rep[2];
    meta->status = 0; meta->litSize = out.litPos; meta->pad0 = 0;
}

}  // namespace zhip
