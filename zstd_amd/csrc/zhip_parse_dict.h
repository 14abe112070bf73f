// zhip_parse_dict.h — gfx950 match finder for small records compressed with an attached dictionary (CDict), strategy dfast.
//
// WHAT it computes: exactly the sequences of ZSTD_compressBlock_doubleFast_dictMatchState_generic
// (lib/compress/zstd_double_fast.c:328-547) for one record (one frame, one block) whose working context has the CDict
// attached (zstd_compress.c:2318-2376, records up to the 16 KB attach cut-off).
//
// HOW.  One wavefront per record, same batch scheme as zhip_parse_dfast.h.  What the dictionary changes:
//   * the record's own two tables are small (hashLog <= 15, chainLog <= 14 after the attach-mode adjustment; 2^11 + 2^10
//     entries for 1 KB records) and positions fit 16 bits, so they live in LDS (6 KB for 1 KB records) — a CU holds a
//     few dozen records at once;
//   * the dictionary's tables are read-only and shared by every record: 32-bit "short cache" entries (index << 8 | tag,
//     zstd_compress_internal.h:1399-1417) in global memory that stay L2-resident; a lane only fetches dictionary bytes
//     when the 8-bit tag matches, exactly as the reference does;
//   * matches and repcodes may start in the dictionary and run on into the record (ZSTD_count_2segments, :797): the
//     wave-wide compares below take (pointer, limit) pairs for both sides.
// Index space as in the reference: dictionary byte j has index j + 2, the record's byte i has index P + i with
// P = dictLen + 2 (the attached context continues where the CDict's window ends, so dictIndexDelta = 0).
#pragma once
#include <hip/hip_runtime.h>
#include "zhip_common.h"
#include "zhip_parse.h"
#include "zhip_parse_dfast.h"

namespace zhip {

// the device view of a CDict (built and uploaded once by the host, zhip_lib.hip)
struct ZhipCDictDev {
    const uint8_t* content;     // dictionary content, 16 zero bytes of padding after the end
    uint32_t len;
    uint32_t hashLog, chainLog, minMatch, strategy;   // the CDict's own parameters
    const uint32_t* tabL;       // tagged: fast -> the hash table; dfast -> long table (hashLog)
    const uint32_t* tabS;       // dfast: short table (chainLog)
    uint32_t rep[3];
    uint32_t dictID;
};

__host__ __device__ inline uint32_t dict_lds_bytes(uint32_t hashLog, uint32_t chainLog) { return (2u << hashLog) + (2u << chainLog) + 2u * ZHIP_DF_SCRATCH; }

__device__ __forceinline__ uint64_t uni64(uint64_t v) { return readlane64(v, 0); }

// 8 bytes of p[o .. o+8) where only p[0 .. l) may be touched (l >= 8); bytes past l read as the buffer's last bytes shifted out
__device__ __forceinline__ uint64_t ld64_lim(const uint8_t* p, uint32_t o, uint32_t l)
{
    if (o + 8 <= l) return ld64(p + o);
    uint32_t const back = o + 8 - l;                        // 1..8 when o < l
    return back >= 8 ? 0 : ld64(p + l - 8) >> (8 * back);
}
// common prefix of a[0..la) and b[0..lb), 512 bytes per round (la, lb >= 8 or the side is empty)
__device__ inline uint32_t wave_count_cross(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t const lim = la < lb ? la : lb;
    uint32_t total = 0;
    for (;;) {
        uint32_t const o = total + 8u * lane;
        uint32_t same = 0;
        if (o < lim) {
            uint64_t const x = ld64_lim(a, o, la) ^ ld64_lim(b, o, lb);
            uint32_t const avail = lim - o;
            same = x ? (uint32_t)(__ffsll((long long)x) - 1) >> 3 : 8;
            if (same > avail) same = avail;
        }
        unsigned long long const stop = __ballot(same < 8);
        if (stop) { int const f = first_lane(stop); return total + 8u * (uint32_t)f + __builtin_amdgcn_readlane(same, f); }
        total += 512;
    }
}
// ZSTD_count_2segments: record bytes src[ipPos..n) against dict[dPos..dictLen) and then, if that runs off the dictionary's
// end, against the record's own start (zstd_compress_internal.h:797-815)
__device__ inline uint32_t wave_count_2seg(const uint8_t* src, uint32_t n, uint32_t ipPos, const uint8_t* dict, uint32_t dictLen, uint32_t dPos)
{
    uint32_t const k = wave_count_cross(src + ipPos, n - ipPos, dict + dPos, dictLen - dPos);
    if (dPos + k != dictLen) return k;
    if (ipPos + k >= n) return k;
    return k + wave_count_cross(src + ipPos + k, n - ipPos - k, src, n);
}
// equal bytes walking backwards from a[pa-1] / b[pb-1], at most `limit`
__device__ inline uint32_t wave_count_back_cross(const uint8_t* a, uint32_t pa, const uint8_t* b, uint32_t pb, uint32_t limit)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t total = 0;
    for (;;) {
        uint32_t const i = total + lane;
        bool const stopHere = (i >= limit) || (a[pa - 1 - i] != b[pb - 1 - i]);
        unsigned long long const stop = __ballot(stopHere);
        if (stop) return total + (uint32_t)first_lane(stop);
        total += 64;
    }
}

// Forward and backward extension in ONE round of loads: lanes 0..47 compare 8 bytes each of a[0..la) / b[0..lb) (384 bytes),
// lanes 48..63 one byte each walking back from ba[pa-1] / bb[pb-1] (at most backLim bytes); longer runs fall back to the loops.
__device__ inline void wave_extend_cross(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb,
                                         const uint8_t* ba, uint32_t pa, const uint8_t* bb, uint32_t pb, uint32_t backLim,
                                         uint32_t& fwd, uint32_t& back)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t const lim = la < lb ? la : lb;
    uint32_t same;
    if (lane < 48) {
        uint32_t const o = 8u * lane;
        same = 0;
        if (o < lim) {
            uint64_t const x = ld64_lim(a, o, la) ^ ld64_lim(b, o, lb);
            uint32_t const avail = lim - o;
            same = x ? (uint32_t)(__ffsll((long long)x) - 1) >> 3 : 8;
            if (same > avail) same = avail;
        }
    } else {
        uint32_t const i = lane - 48;
        bool const stopHere = (i >= backLim) || (ba[pa - 1 - i] != bb[pb - 1 - i]);
        same = stopHere ? 0 : 8;
    }
    unsigned long long const stop = __ballot(same < 8);
    unsigned long long const stopF = stop & 0x0000FFFFFFFFFFFFull, stopB = stop >> 48;
    if (stopF) { int const f = first_lane(stopF); fwd = 8u * (uint32_t)f + __builtin_amdgcn_readlane(same, f); }
    else fwd = 384 + wave_count_cross(a + 384, la - 384, b + 384, lb - 384);
    if (stopB) back = (uint32_t)first_lane(stopB);
    else back = 16 + wave_count_back_cross(ba, pa - 16, bb, pb - 16, backLim - 16);
}
// the same for a match whose other side starts in the dictionary at dPos (forward part = ZSTD_count_2segments)
__device__ inline void wave_extend_dict(const uint8_t* src, uint32_t n, uint32_t ipPos, uint32_t backPos, const uint8_t* dict, uint32_t dictLen,
                                        uint32_t dPos, uint32_t dBackPos, uint32_t backLim, uint32_t& fwd, uint32_t& back)
{
    wave_extend_cross(src + ipPos, n - ipPos, dict + dPos, dictLen - dPos, src, backPos, dict, dBackPos, backLim, fwd, back);
    if (dPos + fwd == dictLen && ipPos + fwd < n) fwd += wave_count_cross(src + ipPos + fwd, n - ipPos - fwd, src, n);
}

// Where a record's own tables live: LDS (16-bit entries, the few KB a record needs: about ten records per CU at the 14 KB the ~1.5 KB
// records of a github-users-like batch ask for) or, for the wavefronts that run BESIDE those on a CU whose LDS is full
// (k_parse_dict_g, 62 registers leave room for three times as many), a per-wavefront region of global memory.  The global form has no
// LDS scratch either: lanes that hash alike are found with one ballot per hash bit (wave_hash_group).
template <bool GLOB> struct DictTabPtr { typedef lds_u16* type; };
template <> struct DictTabPtr<true> { typedef uint16_t* type; };
// exact groups (masks of live lanes with equal key; 0 for a lane alone with its key) + whether any exists
template <bool GLOB>
__device__ __forceinline__ unsigned long long dict_groups(uint32_t key, uint32_t bits, bool live, unsigned long long liveMask, lds_u8* scr)
{
    if constexpr (GLOB) {
        unsigned long long g = wave_hash_group(key, bits) & liveMask;
        if (!live || !(g & (g - 1))) g = 0;
        return __ballot(g != 0) ? g : 0ull;
    } else {
        uint32_t const sl = key & (ZHIP_DF_SCRATCH - 1);
        if (live) scr[sl] = (uint8_t)lane_id();
        __builtin_amdgcn_wave_barrier();
        unsigned long long const lose = __ballot(live && scr[sl] != (uint8_t)lane_id());
        __builtin_amdgcn_wave_barrier();
        return lose ? lane_groups(key, lose, liveMask) : 0ull;
    }
}

template <uint32_t MLS, bool GLOB = false>
__device__ inline void parse_dfast_dms_unit(const uint8_t* __restrict__ src, uint32_t n, const ZhipUnit& u, const ZhipCDictDev& cd,
                                            unsigned char* smem, ZhipSeq* seqs, uint8_t* lits, ZhipParse* meta)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t const shL = 32 - u.hashLog, shS = 32 - u.chainLog;
    uint32_t const dShL = 32 - (cd.hashLog + 8), dShS = 32 - (cd.chainLog + 8);
    uint32_t const dictLen = cd.len, P = dictLen + 2;
    const uint8_t* const dict = cd.content;
    FastOut out; out.seqs = seqs; out.lits = lits; out.nbSeq = 0; out.longPos = 0; out.longType = 0;
    out.litPos = 0; out.pendV = 0; out.pendSh = 0; out.pendOff = 0; out.pendLen = 0;
    typedef typename DictTabPtr<GLOB>::type TabP;
    TabP tabL, tabS; lds_u8* scrL = nullptr; lds_u8* scrS = nullptr;                      // entries: position + 1, 0 = empty
    uint32_t const words = ((2u << u.hashLog) + (2u << u.chainLog)) >> 2;
    if constexpr (GLOB) {
        tabL = (uint16_t*)smem; tabS = (uint16_t*)(smem + (2u << u.hashLog));
        uint32_t* const z = (uint32_t*)smem;
        for (uint32_t i = lane; i < words; i += 64) z[i] = 0;
    } else {
        tabL = (lds_u16*)(uintptr_t)smem;
        tabS = (lds_u16*)(uintptr_t)(smem + (2u << u.hashLog));
        scrL = (lds_u8*)(uintptr_t)(smem + (2u << u.hashLog) + (2u << u.chainLog));
        scrS = scrL + ZHIP_DF_SCRATCH;
        lds_u32* const z = (lds_u32*)(uintptr_t)smem;
        for (uint32_t i = lane; i < words; i += 64) z[i] = 0;
    }
    __builtin_amdgcn_wave_barrier();

    uint32_t anchor = 0, off1 = cd.rep[0], off2 = cd.rep[1];

    if (n >= 9) {                            // ip = 0 < ilimit = n - 8
    uint32_t const nm8 = n - 8;
    int32_t const ilimit = (int32_t)nm8;
    uint32_t ip = 0;                         // :373 ip += (dictAndPrefixLength == 0): a dictionary is attached, so no skip
    uint32_t evAvg16 = 12u << 4, kCap = 16;
    uint32_t nbIp = 0xFFFFFFFFu, nbOff1 = 0, nbRv = 0; uint64_t nbBytes = 0;     // the next batch's source bytes, fetched early
    // bytes of the repcode candidate of record position q (index P + q - off): in the dictionary or in the record
    auto rep_ptr = [&](uint32_t q, uint32_t off) -> const uint8_t* { return off > q ? dict + (dictLen + q - off) : src + (q - off); };
    auto rep_ok = [&](uint32_t q, uint32_t off) -> bool { return (uint32_t)((P - 1) - (P + q - off)) >= 3; };   // ZSTD_index_overlap_check
    for (;;) {                                                               // one turn per match
        if ((int32_t)ip >= ilimit) break;                                    // :378 while (ip < ilimit)
        int evKind = 0;                      // 0 none, 1 repcode, 2 long (record), 3 long (dictionary), 4 short
        uint32_t curr = 0, candE = 0, cand1 = 0; bool long1 = false, dlong1 = false, shortDict = false;
        uint32_t dcandE = 0, dcand1 = 0;
        for (;;) {
            // lanes 0..K-1 search p_j = ip + j*step with step = ((p - anchor) >> 8) + 1 recomputed per position (:443);
            // a batch covers positions that share the step of its first one
            uint32_t const step = ((ip - anchor) >> 8) + 1;
            uint32_t const p = ip + lane * step;
            bool const sameStep = ((p - anchor) >> 8) + 1 == step;
            bool const inRange = (int32_t)p < ilimit;
            int K = __popcll(__ballot(sameStep && inRange));                  // >= 1
            if (K > 63) K = 63;
            if (K > (int)kCap) K = (int)kCap;
            unsigned long long const liveMask = below_mask(K + 1), searchMask = below_mask(K);
            bool const live = (int)lane <= K;                                // lane K = helper for position p_K (ip+1 of lane K-1 when step == 1)

            uint32_t const pc = p < nm8 ? p : nm8;
            bool const pre = nbIp == ip && nbOff1 == off1 && step == 1;      // loaded while the previous sequence was finished
            uint64_t const bytes = pre ? nbBytes : ld64(src + pc);
            uint32_t const vL = mulhi64_top32(bytes, 0xCF1BBCDCB7A56463ULL);
            uint32_t const hl = vL >> shL, hs = hash_pos<MLS>(bytes, shS);
            uint32_t const dHTL = vL >> dShL, dHTS = hash_pos<MLS>(bytes, dShS);
            uint32_t const oldL = live ? (uint32_t)tabL[hl] : 0, oldS = live ? (uint32_t)tabS[hs] : 0;
            uint32_t const dEL = live ? cd.tabL[dHTL >> 8] : 0, dES = live ? cd.tabS[dHTS >> 8] : 0;
            uint32_t const rv = pre ? nbRv : ld32(rep_ptr(pc + 1, off1));
            // record-side candidates (position + 1), with the inserts of earlier lanes of this batch
            uint32_t candL = oldL, candS = oldS;
            uint64_t cbL = oldL ? ld64(src + (oldL - 1)) : ~bytes;
            uint32_t cbS = oldS >= 2 ? ld32(src + (oldS - 1)) : ~(uint32_t)bytes;
            unsigned long long const grpL = dict_groups<GLOB>(hl, u.hashLog, live, liveMask, scrL);
            unsigned long long const grpS = dict_groups<GLOB>(hs, u.chainLog, live, liveMask, scrS);
            unsigned long long const loseL = grpL, loseS = grpS;                 // (a lane's own group: nonzero = it has company)
            if (__ballot(loseL != 0)) {
                unsigned long long const prev = grpL & below_mask((int)lane);
                uint32_t const pd = prev ? 63u - (uint32_t)__clzll((long long)prev) : lane;
                uint32_t const dp = __shfl(p, (int)pd), dlo = __shfl((uint32_t)bytes, (int)pd), dhi = __shfl((uint32_t)(bytes >> 32), (int)pd);
                if (prev) { candL = dp + 1; cbL = ((uint64_t)dhi << 32) | dlo; }
            }
            if (__ballot(loseS != 0)) {
                unsigned long long const prev = grpS & below_mask((int)lane);
                uint32_t const pd = prev ? 63u - (uint32_t)__clzll((long long)prev) : lane;
                uint32_t const dp = __shfl(p, (int)pd), dlo = __shfl((uint32_t)bytes, (int)pd);
                if (prev) { candS = dp + 1; cbS = dlo; }
            }
            // dictionary-side candidates: bytes only where the tag matches (:384-385)
            uint32_t const dIdxL = dEL >> 8, dIdxS = dES >> 8;
            bool const tagL = live && (dEL & 0xFF) == (dHTL & 0xFF), tagS = live && (dES & 0xFF) == (dHTS & 0xFF);
            uint64_t dbL = ~bytes; uint32_t dbS = ~(uint32_t)bytes;
            if (tagL && dIdxL > 2) dbL = ld64(dict + (dIdxL - 2));
            if (tagS && dIdxS > 2) dbS = ld32(dict + (dIdxS - 2));

            bool const hitR = rep_ok(p + 1, off1) && rv == (uint32_t)(bytes >> 8);              // :398
            bool const hitL = candL != 0 && cbL == bytes;                                        // :407 matchIndexL >= prefixLowestIndex
            bool const hitDL = !hitL && tagL && dIdxL > 2 && dbL == bytes;                       // :413
            bool const locS = candS >= 2;                                                        // :427 matchIndexS > prefixLowestIndex
            bool const hitS = locS ? cbS == (uint32_t)bytes : (tagS && dIdxS > 2 && dbS == (uint32_t)bytes);
            unsigned long long const mL = __ballot(hitL), mDL = __ballot(hitDL);
            unsigned long long const mR = __ballot(hitR) & searchMask, mS = __ballot(hitS) & searchMask;
            unsigned long long const mAny = ((mL | mDL) & searchMask) | mR | mS;
            int const jE = mAny ? first_lane(mAny) : 64;
            int const Lcommit = jE < 64 ? jE + 1 : K;
            if (jE < 64) evKind = ((mR >> jE) & 1) ? 1 : (((mL >> jE) & 1) ? 2 : (((mDL >> jE) & 1) ? 3 : 4));
            // :395 hashLong[h2] = hashSmall[h] = curr for every position up to the event: last lane of a group wins
            {   bool const inC = (int)lane < Lcommit;
                unsigned long long const cm = below_mask(Lcommit) & ~below_mask((int)lane + 1);
                if (inC && (grpL & cm) == 0) tabL[hl] = (uint16_t)(p + 1);
                if (inC && (grpS & cm) == 0) tabS[hs] = (uint16_t)(p + 1);
            }
            __builtin_amdgcn_wave_barrier();
            if (evKind) {
                evAvg16 = (3 * evAvg16 + (((uint32_t)jE + 1) << 4)) >> 2;
                kCap = (evAvg16 >> 4) + 4; if (kCap > 63) kCap = 63;
                curr = ip + (uint32_t)jE * step;
                candE = __builtin_amdgcn_readlane(evKind == 2 ? candL : candS, jE);
                dcandE = __builtin_amdgcn_readlane(evKind == 3 ? dIdxL : dIdxS, jE);
                shortDict = evKind == 4 && !((__ballot(locS) >> jE) & 1);
                if (evKind == 4) {
                    // :447-456 _search_next_long looks at position curr+1 and always inserts it into the long table.
                    // With step == 1 that position is lane jE+1 (live: jE < K); otherwise it is fetched here.
                    if (step == 1) {
                        cand1 = __builtin_amdgcn_readlane(candL, jE + 1);
                        long1 = (mL >> (jE + 1)) & 1;
                        dcand1 = __builtin_amdgcn_readlane(dIdxL, jE + 1);
                        dlong1 = (mDL >> (jE + 1)) & 1;
                        if ((int)lane == jE + 1) tabL[hl] = (uint16_t)(p + 1);
                    } else {
                        uint32_t const q = curr + 1;                                             // q <= nm8: curr < ilimit
                        uint64_t const b1 = ld64(src + q);
                        uint32_t const v1 = mulhi64_top32(b1, 0xCF1BBCDCB7A56463ULL);
                        uint32_t const o1 = (uint32_t)tabL[v1 >> shL];
                        uint32_t const e1 = cd.tabL[(v1 >> dShL) >> 8];
                        long1 = o1 != 0 && ld64(src + (o1 ? o1 - 1 : 0)) == b1;
                        cand1 = o1;
                        dcand1 = e1 >> 8;
                        dlong1 = !long1 && (e1 & 0xFF) == ((v1 >> dShL) & 0xFF) && dcand1 > 2 && ld64(dict + (dcand1 > 2 ? dcand1 - 2 : 0)) == b1;
                        __builtin_amdgcn_wave_barrier();
                        if (lane == 0) tabL[v1 >> shL] = (uint16_t)(q + 1);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                break;
            }
            ip = ip + (uint32_t)K * step;                                    // :443, K positions without a match
            kCap = kCap * 2 > 63 ? 63 : kCap * 2;
            if ((int32_t)ip >= ilimit) break;
        }
        if (evKind == 0) break;

        uint32_t mLength, offBase, mstart = curr;
        if (evKind == 1) {                                                   // :398-405
            mstart = curr + 1;
            uint32_t const q = mstart + 4;
            if (off1 > mstart) {                                             // the repcode points into the dictionary
                uint32_t const dPos = dictLen + mstart - off1;
                mLength = 4 + wave_count_2seg(src, n, q, dict, dictLen, dPos + 4);
            } else mLength = 4 + wave_count_cross(src + q, n - q, src + (q - off1), n - (q - off1));
            offBase = 1;
        } else {
            uint32_t offset;
            if (evKind == 2) {                                               // :407-412 record long match
                uint32_t const m = candE - 1;
                offset = curr - m;
                uint32_t const lim = (curr - anchor) < m ? (curr - anchor) : m;
                uint32_t fwd, back;
                wave_extend_cross(src + curr + 8, n - curr - 8, src + m + 8, n - m - 8, src, curr, src, m, lim, fwd, back);
                mstart = curr - back; mLength = 8 + fwd + back;
            } else if (evKind == 3) {                                        // :413-424 dictionary long match
                uint32_t const dPos = dcandE - 2;
                offset = (P + curr) - dcandE;
                uint32_t const lim = (curr - anchor) < dPos ? (curr - anchor) : dPos;           // dm > dictStart
                uint32_t fwd, back;
                wave_extend_dict(src, n, curr + 8, curr, dict, dictLen, dPos + 8, dPos, lim, fwd, back);
                mstart = curr - back; mLength = 8 + fwd + back;
            } else if (long1) {                                              // :459-464 record long match at curr + 1
                uint32_t const q = curr + 1, m = cand1 - 1;
                offset = q - m;
                uint32_t const lim = (q - anchor) < m ? (q - anchor) : m;
                uint32_t fwd, back;
                wave_extend_cross(src + q + 8, n - q - 8, src + m + 8, n - m - 8, src, q, src, m, lim, fwd, back);
                mstart = q - back; mLength = 8 + fwd + back;
            } else if (dlong1) {                                             // :465-477 dictionary long match at curr + 1
                uint32_t const q = curr + 1, dPos = dcand1 - 2;
                offset = (P + q) - dcand1;
                uint32_t const lim = (q - anchor) < dPos ? (q - anchor) : dPos;
                uint32_t fwd, back;
                wave_extend_dict(src, n, q + 8, q, dict, dictLen, dPos + 8, dPos, lim, fwd, back);
                mstart = q - back; mLength = 8 + fwd + back;
            } else if (shortDict) {                                          // :481-484 the short match lies in the dictionary
                uint32_t const dPos = dcandE - 2;
                offset = (P + curr) - dcandE;
                uint32_t const lim = (curr - anchor) < dPos ? (curr - anchor) : dPos;
                uint32_t fwd, back;
                wave_extend_dict(src, n, curr + 4, curr, dict, dictLen, dPos + 4, dPos, lim, fwd, back);
                mstart = curr - back; mLength = 4 + fwd + back;
            } else {                                                         // :485-489
                uint32_t const m = candE - 1;
                offset = curr - m;
                uint32_t const lim = (curr - anchor) < m ? (curr - anchor) : m;
                uint32_t fwd, back;
                wave_extend_cross(src + curr + 4, n - curr - 4, src + m + 4, n - m - 4, src, curr, src, m, lim, fwd, back);
                mstart = curr - back; mLength = 4 + fwd + back;
            }
            off2 = off1; off1 = offset;
            offBase = offset + 3;
        }
        lits_copy(out, src, nm8, anchor, mstart - anchor);
        store_seq(out, mstart - anchor, offBase, mLength);
        ip = mstart + mLength; anchor = ip;

        if ((int32_t)ip <= ilimit) {                                         // :503-535
            // ONE round of loads: the bytes of the four complementary inserts, the immediate-repcode probe and the next batch
            uint32_t const q = lane == 0 ? curr + 2 : (lane == 1 ? ip - 2 : ip - 1);
            uint64_t const bq = ld64(src + (q < nm8 ? q : nm8));
            uint64_t bIp = uni64(ld64(src + ip));
            bool r2ok = rep_ok(ip, off2);
            uint32_t r2 = r2ok ? uni(ld32(rep_ptr(ip, off2))) : 0;
            {   uint32_t const np = ip + lane, npc = np < nm8 ? np : nm8;
                nbBytes = ld64(src + npc); nbRv = ld32(rep_ptr(npc + 1, off1)); nbIp = ip; nbOff1 = off1;
            }
            {   // complementary inserts: long[curr+2], long[ip-2], short[curr+2], short[ip-1] — in this order
                uint32_t const hL = mulhi64_top32(bq, 0xCF1BBCDCB7A56463ULL) >> shL, hS = hash_pos<MLS>(bq, shS);
                if (lane == 0) { tabL[hL] = (uint16_t)(q + 1); tabS[hS] = (uint16_t)(q + 1); }
                __builtin_amdgcn_wave_barrier();
                if (lane == 1) tabL[hL] = (uint16_t)(q + 1);
                if (lane == 2) tabS[hS] = (uint16_t)(q + 1);
                __builtin_amdgcn_wave_barrier();
            }
            while ((int32_t)ip <= ilimit) {
                if (!r2ok || (uint32_t)bIp != r2) break;
                uint32_t rl;
                if (off2 > ip) rl = 4 + wave_count_2seg(src, n, ip + 4, dict, dictLen, dictLen + ip - off2 + 4);
                else rl = 4 + wave_count_cross(src + ip + 4, n - ip - 4, src + (ip + 4 - off2), n - (ip + 4 - off2));
                {   uint32_t const t = off2; off2 = off1; off1 = t; }
                if (lane == 0) { tabS[hash_pos<MLS>(bIp, shS)] = (uint16_t)(ip + 1); tabL[mulhi64_top32(bIp, 0xCF1BBCDCB7A56463ULL) >> shL] = (uint16_t)(ip + 1); }
                __builtin_amdgcn_wave_barrier();
                store_seq(out, 0, 1, rl);
                ip += rl; anchor = ip;
                if ((int32_t)ip > ilimit) break;
                bIp = uni64(ld64(src + ip));
                r2ok = rep_ok(ip, off2);
                r2 = r2ok ? uni(ld32(rep_ptr(ip, off2))) : 0;
            }
        }
    }
    lits_copy(out, src, nm8, anchor, n - anchor);                           // trailing literals
    lits_flush(out);
    } else {
        for (uint32_t i = lane; i < n; i += 64) lits[i] = src[i];
        out.litPos = n;
    }
    if (lane == 0) {
        meta->nbSeq = out.nbSeq; meta->lastLits = n - anchor;
        meta->longPos = out.longPos; meta->longType = out.longType;
        meta->rep[0] = off1; meta->rep[1] = off2; meta->rep[2] = cd.rep[2];
        meta->status = 0; meta->litSize = out.litPos; meta->pad0 = 0;
    }
}

// ------------------------------------------------------------------ strategy fast with an attached dictionary
// ZSTD_compressBlock_fast_dictMatchState_generic (lib/compress/zstd_fast.c:483-678).  One table for the record (LDS, 16-bit),
// one tagged table of the dictionary; per position, in the reference's order: repcode at p+1 (:566), dictionary match — used
// only when the record's own entry is invalid (:575-596) — then the record's own match (:598).  Positions advance like the
// noDict dfast loop (step grows every 256 bytes without a match, :618-625), so the batching is the one of zhip_parse_dfast.h.
__host__ __device__ inline uint32_t dict_fast_lds_bytes(uint32_t hashLog) { return (2u << hashLog) + ZHIP_DF_SCRATCH; }

template <uint32_t MLS, bool GLOB = false>
__device__ inline void parse_fast_dms_unit(const uint8_t* __restrict__ src, uint32_t n, const ZhipUnit& u, const ZhipCDictDev& cd,
                                           unsigned char* smem, ZhipSeq* seqs, uint8_t* lits, ZhipParse* meta)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t const sh = 32 - u.hashLog, dSh = 32 - (cd.hashLog + 8);
    uint32_t const stepSize = u.targetLength + !u.targetLength;                 // :491
    uint32_t const dictLen = cd.len, P = dictLen + 2;
    const uint8_t* const dict = cd.content;
    FastOut out; out.seqs = seqs; out.lits = lits; out.nbSeq = 0; out.longPos = 0; out.longType = 0;
    out.litPos = 0; out.pendV = 0; out.pendSh = 0; out.pendOff = 0; out.pendLen = 0;
    typedef typename DictTabPtr<GLOB>::type TabP;
    TabP tab; lds_u8* scr = nullptr;                                                      // entries: position + 1, 0 = empty
    uint32_t const words = (2u << u.hashLog) >> 2;
    if constexpr (GLOB) {
        tab = (uint16_t*)smem;
        uint32_t* const z = (uint32_t*)smem;
        for (uint32_t i = lane; i < words; i += 64) z[i] = 0;
    } else {
        tab = (lds_u16*)(uintptr_t)smem;
        scr = (lds_u8*)(uintptr_t)(smem + (2u << u.hashLog));
        lds_u32* const z = (lds_u32*)(uintptr_t)smem;
        for (uint32_t i = lane; i < words; i += 64) z[i] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t anchor = 0, off1 = cd.rep[0], off2 = cd.rep[1];

    if (n >= 8 + stepSize) {                 // ip1 = stepSize <= ilimit = n - 8
    uint32_t const nm8 = n - 8;
    int32_t const ilimit = (int32_t)nm8;
    uint32_t ip = 0;
    uint32_t evAvg16 = 12u << 4, kCap = 16;
    auto rep_ptr = [&](uint32_t q, uint32_t off) -> const uint8_t* { return off > q ? dict + (dictLen + q - off) : src + (q - off); };
    auto rep_ok = [&](uint32_t q, uint32_t off) -> bool { return (uint32_t)((P - 1) - (P + q - off)) >= 3; };
    for (;;) {                                                               // one turn per match (:537 while (ip1 <= ilimit))
        uint32_t step = stepSize, nextStep = ip + 256;
        if ((int32_t)(ip + stepSize) > ilimit) break;
        int evKind = 0;                      // 0 none, 1 repcode, 2 dictionary match, 3 record match
        uint32_t curr = 0, candE = 0;
        for (;;) {
            uint32_t const p = ip + lane * step;
            bool const inc = (int32_t)(p + step) >= (int32_t)nextStep;                    // :618 step++ after this position
            bool const endAfter = (int32_t)(p + 2 * step + (inc ? 1u : 0u)) > ilimit;     // :624 the next position does not run
            unsigned long long const mStop = __ballot(inc || endAfter);
            int K = mStop ? first_lane(mStop) + 1 : 64;
            if (K > (int)kCap) K = (int)kCap;
            bool const lastInc = (__ballot(inc) >> (K - 1)) & 1, lastEnd = (__ballot(endAfter) >> (K - 1)) & 1;
            unsigned long long const liveMask = below_mask(K);
            bool const live = (int)lane < K;

            uint32_t const pc = p < nm8 ? p : nm8;
            uint64_t const bytes = ld64(src + pc);
            uint32_t const h = hash_pos<MLS>(bytes, sh), dHT = hash_pos<MLS>(bytes, dSh);
            uint32_t const old = live ? (uint32_t)tab[h] : 0;
            uint32_t const dE = live ? cd.tabL[dHT >> 8] : 0;
            uint32_t const rv = ld32(rep_ptr(pc + 1, off1));
            uint32_t cand = old;
            uint32_t cb = old ? ld32(src + (old - 1)) : ~(uint32_t)bytes;
            unsigned long long const grp = dict_groups<GLOB>(h, u.hashLog, live, liveMask, scr);
            if (__ballot(grp != 0)) {
                unsigned long long const prev = grp & below_mask((int)lane);
                uint32_t const pd = prev ? 63u - (uint32_t)__clzll((long long)prev) : lane;
                uint32_t const dp = __shfl(p, (int)pd), dlo = __shfl((uint32_t)bytes, (int)pd);
                if (prev) { cand = dp + 1; cb = dlo; }
            }
            uint32_t const dIdx = dE >> 8;
            bool const tagOk = live && (dE & 0xFF) == (dHT & 0xFF);
            uint32_t db = ~(uint32_t)bytes;
            if (tagOk && dIdx > 2) db = ld32(dict + (dIdx - 2));
            bool const hitR = live && rep_ok(p + 1, off1) && rv == (uint32_t)(bytes >> 8);                 // :566
            bool const hitD = tagOk && dIdx > 2 && db == (uint32_t)bytes && cand <= 1;                      // :575-583 matchIndex <= prefixStartIndex
            bool const hitM = live && cand != 0 && cb == (uint32_t)bytes;                                   // :598 ZSTD_match4Found_cmov
            unsigned long long const mR = __ballot(hitR), mD = __ballot(hitD), mM = __ballot(hitM);
            unsigned long long const mAny = mR | mD | mM;
            int const jE = mAny ? first_lane(mAny) : 64;
            int const Lcommit = jE < 64 ? jE + 1 : K;
            if (jE < 64) evKind = ((mR >> jE) & 1) ? 1 : (((mD >> jE) & 1) ? 2 : 3);
            {   bool const inC = (int)lane < Lcommit;                        // :564 hashTable[hash0] = curr up to and including the event
                unsigned long long const cm = below_mask(Lcommit) & ~below_mask((int)lane + 1);
                if (inC && (grp & cm) == 0) tab[h] = (uint16_t)(p + 1);
            }
            __builtin_amdgcn_wave_barrier();
            if (evKind) {
                evAvg16 = (3 * evAvg16 + (((uint32_t)jE + 1) << 4)) >> 2;
                kCap = (evAvg16 >> 4) + 4; if (kCap > 64) kCap = 64;
                curr = ip + (uint32_t)jE * step;
                candE = __builtin_amdgcn_readlane(evKind == 2 ? dIdx : cand, jE);
                break;
            }
            ip = ip + (uint32_t)K * step;
            kCap = kCap * 2 > 64 ? 64 : kCap * 2;
            if (lastEnd) break;
            if (lastInc) { step++; nextStep += 256; }
        }
        if (evKind == 0) break;

        uint32_t mLength, offBase, mstart = curr;
        if (evKind == 1) {                                                   // :566-573
            mstart = curr + 1;
            uint32_t const q = mstart + 4;
            if (off1 > mstart) mLength = 4 + wave_count_2seg(src, n, q, dict, dictLen, dictLen + mstart - off1 + 4);
            else mLength = 4 + wave_count_cross(src + q, n - q, src + (q - off1), n - (q - off1));
            offBase = 1;
        } else {
            uint32_t offset;
            if (evKind == 2) {                                               // :584-595 dictionary match
                uint32_t const dPos = candE - 2;
                mLength = 4 + wave_count_2seg(src, n, curr + 4, dict, dictLen, dPos + 4);
                offset = (P + curr) - candE;
                uint32_t const lim = (curr - anchor) < dPos ? (curr - anchor) : dPos;
                uint32_t const back = wave_count_back_cross(src, curr, dict, dPos, lim);
                mstart = curr - back; mLength += back;
            } else {                                                         // :598-611
                uint32_t const m = candE - 1;
                mLength = 4 + wave_count_cross(src + curr + 4, n - curr - 4, src + m + 4, n - m - 4);
                offset = curr - m;
                uint32_t const lim = (curr - anchor) < m ? (curr - anchor) : m;
                uint32_t const back = wave_count_back(src, curr, m, lim);
                mstart = curr - back; mLength += back;
            }
            off2 = off1; off1 = offset;
            offBase = offset + 3;
        }
        lits_copy(out, src, nm8, anchor, mstart - anchor);
        store_seq(out, mstart - anchor, offBase, mLength);
        ip = mstart + mLength; anchor = ip;

        if ((int32_t)ip <= ilimit) {                                         // :635-663
            {   uint32_t const q = lane == 0 ? curr + 2 : ip - 2;
                uint64_t const b = ld64(src + (q < nm8 ? q : nm8));
                uint32_t const hh = hash_pos<MLS>(b, sh);
                if (lane == 0) tab[hh] = (uint16_t)(q + 1);
                __builtin_amdgcn_wave_barrier();
                if (lane == 1) tab[hh] = (uint16_t)(q + 1);
                __builtin_amdgcn_wave_barrier();
            }
            while ((int32_t)ip <= ilimit) {
                uint64_t const b = ld64(src + ip);
                if (!rep_ok(ip, off2) || (uint32_t)b != uni(ld32(rep_ptr(ip, off2)))) break;
                uint32_t rl;
                if (off2 > ip) rl = 4 + wave_count_2seg(src, n, ip + 4, dict, dictLen, dictLen + ip - off2 + 4);
                else rl = 4 + wave_count_cross(src + ip + 4, n - ip - 4, src + (ip + 4 - off2), n - (ip + 4 - off2));
                {   uint32_t const t = off2; off2 = off1; off1 = t; }
                if (lane == 0) tab[hash_pos<MLS>(b, sh)] = (uint16_t)(ip + 1);
                __builtin_amdgcn_wave_barrier();
                store_seq(out, 0, 1, rl);
                ip += rl; anchor = ip;
            }
        }
    }
    lits_copy(out, src, nm8, anchor, n - anchor);
    lits_flush(out);
    } else {
        for (uint32_t i = lane; i < n; i += 64) lits[i] = src[i];
        out.litPos = n;
    }
    if (lane == 0) {
        meta->nbSeq = out.nbSeq; meta->lastLits = n - anchor;
        meta->longPos = out.longPos; meta->longType = out.longType;
        meta->rep[0] = off1; meta->rep[1] = off2; meta->rep[2] = cd.rep[2];
        meta->status = 0; meta->litSize = out.litPos; meta->pad0 = 0;
    }
}

}  // namespace zhip
