/* oracle/zoracle.c — TEST INFRASTRUCTURE ONLY (see zoracle.h).
 *
 * A deliberately plain, serial restatement of the reference's hot path.  Every function cites the reference
 * lines it follows (paths relative to /root/reference).  Parity status: PINNED — byte-identical to the real
 * reference (oracle/_ref) on the sweeps in tests/test_oracle_vs_reference.py and on tests/golden/ fixtures.
 *
 * Style notes: positions are 0-based offsets into the block (the reference uses window indices = pos + base
 * offset; only differences ever reach the output).  Hash-table value 0 means "empty"; stored
 * values are pos+1.
 */
#include "zoracle.h"
#include <string.h>
#include <stdlib.h>

/* ------------------------------------------------------------------ small helpers */
static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint32_t rd16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
static unsigned hb32(uint32_t v) { return 31u - (unsigned)__builtin_clz(v); }          /* lib/common/bits.h:177 */
static void wr16(uint8_t* p, unsigned v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void wr24(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); }
static void wr32(uint8_t* p, uint32_t v) { wr16(p, v & 0xFFFF); wr16(p + 2, v >> 16); }

/* ------------------------------------------------------------------ parameters */
/* lib/compress/clevels.h:24-130, rows "base for negative levels" .. level 12 of the four size classes
 * (strategy: 1 fast, 2 dfast, 3 greedy, 4 lazy, 5 lazy2; 6+ = binary-tree strategies, outside this oracle) */
static const zo_cparams kRows[4][13] = {
  { {19,12,13,1,6,1,1}, {19,13,14,1,7,0,1}, {20,15,16,1,6,0,1}, {21,16,17,1,5,0,2}, {21,18,18,1,5,0,2},      /* > 256 KB  */
    {21,18,19,3,5,2,3}, {21,18,19,3,5,4,4}, {21,19,20,4,5,8,4}, {21,19,20,4,5,16,5}, {22,20,21,4,5,16,5},
    {22,21,22,5,5,16,5}, {22,21,22,6,5,16,5}, {22,22,23,6,5,32,5} },
  { {18,12,13,1,5,1,1}, {18,13,14,1,6,0,1}, {18,14,14,1,5,0,2}, {18,16,16,1,4,0,2}, {18,16,17,3,5,2,3},      /* <= 256 KB */
    {18,17,18,5,5,2,3}, {18,18,19,3,5,4,4}, {18,18,19,4,4,4,4}, {18,18,19,4,4,8,5}, {18,18,19,5,4,8,5},
    {18,18,19,6,4,8,5}, {18,18,19,5,4,12,6}, {18,19,19,7,4,12,6} },
  { {17,12,12,1,5,1,1}, {17,12,13,1,6,0,1}, {17,13,15,1,5,0,1}, {17,15,16,2,5,0,2}, {17,17,17,2,4,0,2},      /* <= 128 KB */
    {17,16,17,3,4,2,3}, {17,16,17,3,4,4,4}, {17,16,17,3,4,8,5}, {17,16,17,4,4,8,5}, {17,16,17,5,4,8,5},
    {17,16,17,6,4,8,5}, {17,17,17,5,4,8,6}, {17,18,17,7,4,12,6} },
  { {14,12,13,1,5,1,1}, {14,14,15,1,5,0,1}, {14,14,15,1,4,0,1}, {14,14,15,2,4,0,2}, {14,14,14,4,4,2,3},      /* <= 16 KB  */
    {14,14,14,3,4,4,4}, {14,14,14,4,4,8,5}, {14,14,14,6,4,8,5}, {14,14,14,8,4,8,5}, {14,15,14,5,4,8,6},
    {14,15,14,9,4,8,6}, {14,15,14,3,4,12,7}, {14,15,14,4,3,24,7} },
};

/* zstd_compress.c:1466-1602 ZSTD_adjustCParams_internal.  mode: 0 = noAttachDict / unknown, 1 = attachDict, 2 = createCDict */
#define ZO_SRCSIZE_UNKNOWN (~0ULL)
static void zo_adjust_cparams(zo_cparams* cp, unsigned long long srcSize, unsigned long long dictSize, int mode)
{
    if (mode == 2 && dictSize && srcSize == ZO_SRCSIZE_UNKNOWN) srcSize = 513;   /* :1524-1531 minSrcSize */
    if (mode == 1) dictSize = 0;                                                 /* :1532-1538 the dictionary has its own tables */
    if (srcSize <= (1ULL << 30) && dictSize <= (1ULL << 30)) {                   /* :1546-1553 */
        uint32_t const tSize = (uint32_t)(srcSize + dictSize);
        unsigned const srcLog = (tSize < 64) ? 6 : hb32(tSize - 1) + 1;
        if (cp->windowLog > srcLog) cp->windowLog = srcLog;
    }
    if (srcSize != ZO_SRCSIZE_UNKNOWN) {                                         /* :1554-1560 */
        unsigned dawl = cp->windowLog;                                           /* :1432-1456 ZSTD_dictAndWindowLog */
        if (dictSize) {
            unsigned long long const windowSize = 1ULL << cp->windowLog;
            if (windowSize < dictSize + srcSize) dawl = (dictSize + windowSize >= (1ULL << 31)) ? 31 : hb32((uint32_t)(dictSize + windowSize) - 1) + 1;
        }
        if (cp->hashLog > dawl + 1) cp->hashLog = dawl + 1;
        if (cp->chainLog > dawl) cp->chainLog = dawl;                            /* cycleLog == chainLog below btlazy2 */
    }
    if (cp->windowLog < 10) cp->windowLog = 10;                                  /* :1562 */
    if (mode == 2 && cp->strategy <= 2) {                                        /* :1568-1576 tagged CDict tables: 24 bits of hash at most */
        if (cp->hashLog > 24) cp->hashLog = 24;
        if (cp->chainLog > 24) cp->chainLog = 24;
    }
}

/* zstd_compress.c:7098-7145 ZSTD_getCParamRowSize + ZSTD_getCParams_internal */
static int g_zo_any_strategy = 0;   /* 1: do not refuse rows above lazy2 (a caller that only wants the row's windowLog) */
static int zo_get_cparams_mode(int level, unsigned long long srcSize, unsigned long long dictSize, int mode, zo_cparams* out)
{
    unsigned long long const rowDict = (mode == 1) ? 0 : dictSize;
    int const unknown = srcSize == ZO_SRCSIZE_UNKNOWN;
    unsigned long long const rSize = (unknown && rowDict == 0) ? ZO_SRCSIZE_UNKNOWN : srcSize + rowDict + ((unknown && rowDict > 0) ? 500 : 0);
    unsigned const tableID = (rSize <= 256u*1024) + (rSize <= 128u*1024) + (rSize <= 16u*1024);
    int row = level;
    zo_cparams cp;
    if (level == 0) row = 3;                       /* ZSTD_CLEVEL_DEFAULT */
    if (level < 0) row = 0;
    if (row > 12) return -1;                       /* only binary-tree strategies up there */
    cp = kRows[tableID][row];
    if (level < 0) {                               /* :7139-7142, ZSTD_minCLevel() = -(1<<17) */
        int const clamped = level < -131072 ? -131072 : level;
        cp.targetLength = (unsigned)(-clamped);
    }
    if (cp.strategy > 5 && !g_zo_any_strategy) return -1;   /* btlazy2 and up are outside this oracle */
    zo_adjust_cparams(&cp, srcSize, dictSize, mode);
    *out = cp;
    return 0;
}

/* no dictionary, known srcSize */
int zo_get_cparams(int level, unsigned long long srcSize, zo_cparams* out)
{
    return zo_get_cparams_mode(level, srcSize, 0, 0, out);
}

size_t zo_compress_bound(size_t n)   /* lib/zstd.h:235 */
{
    return n + (n >> 8) + ((n < (128u << 10)) ? (((128u << 10) - n) >> 11) : 0);
}

/* ------------------------------------------------------------------ stage 1: match finders */
/* zstd_compress_internal.h:820-862 */
static uint32_t zo_hash(const uint8_t* p, unsigned hBits, unsigned mls)
{
    switch (mls) {
    default:
    case 4: return (rd32(p) * 2654435761U) >> (32 - hBits);
    case 5: return (uint32_t)(((rd64(p) << 24) * 889523592379ULL) >> (64 - hBits));
    case 6: return (uint32_t)(((rd64(p) << 16) * 227718039650203ULL) >> (64 - hBits));
    case 7: return (uint32_t)(((rd64(p) << 8) * 58295818150454627ULL) >> (64 - hBits));
    case 8: return (uint32_t)((rd64(p) * 0xCF1BBCDCB7A56463ULL) >> (64 - hBits));
    }
}

/* zstd_compress_internal.h:771 — common prefix length of src[a..) and src[b..), a bounded by n */
static uint32_t zo_count(const uint8_t* src, size_t a, size_t b, size_t n)
{
    size_t const a0 = a;
    while (a < n && src[a] == src[b]) { a++; b++; }
    return (uint32_t)(a - a0);
}

typedef struct {
    zo_seq* seqs; size_t nb, cap;
    uint8_t* lits; size_t litSize;
    int overflow;
} zo_store;

/* zstd_compress_internal.h:671 (literal copy + one seqDef) */
static void zo_store_seq(zo_store* st, const uint8_t* src, size_t anchor, size_t litLength, uint32_t offBase, uint32_t ml)
{
    if (st->nb >= st->cap) { st->overflow = 1; return; }
    memcpy(st->lits + st->litSize, src + anchor, litLength);
    st->litSize += litLength;
    st->seqs[st->nb].litLength = (uint32_t)litLength;
    st->seqs[st->nb].matchLength = ml;
    st->seqs[st->nb].offBase = offBase;
    st->nb++;
}

/* zstd_fast.c:192-423  ZSTD_compressBlock_fast_noDict_generic, one block, empty history.
 * T[] holds pos+1 (0 = empty).  Returns trailing-literal count; rep[] updated like :368-372. */
static size_t zo_fast(const zo_cparams* cp, const uint8_t* src, size_t n, zo_store* st, uint32_t rep[3])
{
    unsigned const hlog = cp->hashLog, mls = cp->minMatch;
    size_t const stepSize = cp->targetLength + !cp->targetLength + 1;            /* :200 */
    uint32_t* T = (uint32_t*)calloc((size_t)1 << hlog, sizeof(uint32_t));
    size_t const ilimit = n - 8;                                                 /* :207, caller guarantees n >= 8 */
    size_t anchor = 0, ip0 = 1, ip1, ip2, ip3, cur0 = 0, step, nextStep, match0 = 0, mLength;
    uint32_t rep1 = rep[0], rep2 = rep[1], saved1 = 0, saved2 = 0, offBase;
    uint32_t h0, h1, cand;                                                       /* cand = candidate pos+1 for ip0 */
    {   uint32_t const maxRep = 1;                                               /* :238-244: ip0 = 1, window low = 0 */
        if (rep2 > maxRep) { saved2 = rep2; rep2 = 0; }
        if (rep1 > maxRep) { saved1 = rep1; rep1 = 0; }
    }
    for (;;) {   /* _start */
        step = stepSize; nextStep = ip0 + 128;                                   /* :249-250, kStepIncr = 1<<7 */
        ip1 = ip0 + 1; ip2 = ip0 + step; ip3 = ip2 + 1;
        if (ip3 >= ilimit) break;                                                /* :257 */
        h0 = zo_hash(src + ip0, hlog, mls); h1 = zo_hash(src + ip1, hlog, mls);
        cand = T[h0];
        for (;;) {
            int found = 0;
            uint32_t const rval = rep1 ? rd32(src + ip2 - rep1) : 0;
            cur0 = ip0; T[h0] = (uint32_t)ip0 + 1;                               /* :271-272 */
            if (rep1 > 0 && rd32(src + ip2) == rval) {                           /* :275-290 repcode at ip2 */
                ip0 = ip2; match0 = ip0 - rep1;
                mLength = (src[ip0 - 1] == src[match0 - 1]);
                ip0 -= mLength; match0 -= mLength;
                offBase = 1; mLength += 4;
                T[h1] = (uint32_t)ip1 + 1;
                found = 2;
            } else if (cand && rd32(src + ip0) == rd32(src + cand - 1)) {        /* :292-299 */
                T[h1] = (uint32_t)ip1 + 1;
                found = 1;
            } else {
                cand = T[h1]; h0 = h1; h1 = zo_hash(src + ip2, hlog, mls);       /* :302-311 */
                ip0 = ip1; ip1 = ip2; ip2 = ip3;
                cur0 = ip0; T[h0] = (uint32_t)ip0 + 1;                           /* :314-315 */
                if (cand && rd32(src + ip0) == rd32(src + cand - 1)) {           /* :317-326 */
                    if (step <= 4) T[h1] = (uint32_t)ip1 + 1;
                    found = 1;
                } else {
                    cand = T[h1]; h0 = h1; h1 = zo_hash(src + ip2, hlog, mls);   /* :329-339 */
                    ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step;
                    if (ip2 >= nextStep) { step++; nextStep += 128; }            /* :342-347 */
                    if (ip3 < ilimit) continue;
                    goto cleanup;
                }
            }
            if (found == 1) {   /* _offset :377-391 */
                match0 = cand - 1;
                rep2 = rep1; rep1 = (uint32_t)(ip0 - match0);
                offBase = rep1 + 3; mLength = 4;
                while (ip0 > anchor && match0 > 0 && src[ip0 - 1] == src[match0 - 1]) { ip0--; match0--; mLength++; }
            }
            /* _match :393-401 */
            mLength += zo_count(src, ip0 + mLength, match0 + mLength, n);
            zo_store_seq(st, src, anchor, ip0 - anchor, offBase, (uint32_t)mLength);
            ip0 += mLength; anchor = ip0;
            if (ip0 <= ilimit) {                                                 /* :404-420 */
                T[zo_hash(src + cur0 + 2, hlog, mls)] = (uint32_t)cur0 + 2 + 1;
                T[zo_hash(src + ip0 - 2, hlog, mls)] = (uint32_t)ip0 - 2 + 1;
                if (rep2 > 0) {
                    while (ip0 <= ilimit && rd32(src + ip0) == rd32(src + ip0 - rep2)) {
                        uint32_t const rLength = zo_count(src, ip0 + 4, ip0 + 4 - rep2, n) + 4;
                        uint32_t const t = rep2; rep2 = rep1; rep1 = t;
                        T[zo_hash(src + ip0, hlog, mls)] = (uint32_t)ip0 + 1;
                        ip0 += rLength;
                        zo_store_seq(st, src, anchor, 0, 1, rLength);
                        anchor = ip0;
                    }
                }
            }
            break;   /* goto _start */
        }
    }
cleanup:
    saved2 = (saved1 != 0 && rep1 != 0) ? saved1 : saved2;                       /* :368 */
    rep[0] = rep1 ? rep1 : saved1;
    rep[1] = rep2 ? rep2 : saved2;
    free(T);
    return n - anchor;
}

/* zstd_double_fast.c:105-323  ZSTD_compressBlock_doubleFast_noDict_generic, one block, empty history */
static size_t zo_dfast(const zo_cparams* cp, const uint8_t* src, size_t n, zo_store* st, uint32_t rep[3])
{
    unsigned const hL = cp->hashLog, hS = cp->chainLog, mls = cp->minMatch;
    uint32_t* TL = (uint32_t*)calloc((size_t)1 << hL, sizeof(uint32_t));
    uint32_t* TS = (uint32_t*)calloc((size_t)1 << hS, sizeof(uint32_t));
    size_t const ilimit = n - 8;
    size_t anchor = 0, ip = 1, ip1, step, nextStep, curr = 0, mLength = 0;
    uint32_t off1 = rep[0], off2 = rep[1], saved1 = 0, saved2 = 0, offset = 0;
    {   uint32_t const maxRep = 1;
        if (off2 > maxRep) { saved2 = off2; off2 = 0; }
        if (off1 > maxRep) { saved1 = off1; off1 = 0; }
    }
    for (;;) {
        uint32_t hl0, hl1 = 0, idxl0, idxl1 = 0;      /* idx* are pos+1, 0 = empty */
        int kind = 0;                                  /* 1 = repcode stored, 2 = match found */
        step = 1; nextStep = ip + 256; ip1 = ip + step;                           /* :168-170, kStepIncr = 1<<8 */
        if (ip1 > ilimit) break;
        hl0 = zo_hash(src + ip, hL, 8); idxl0 = TL[hl0];
        do {
            uint32_t const hs0 = zo_hash(src + ip, hS, mls);
            uint32_t const idxs0 = TS[hs0];
            size_t matchs0;
            curr = ip;
            TL[hl0] = TS[hs0] = (uint32_t)ip + 1;                                /* :187 */
            if (off1 > 0 && rd32(src + ip + 1 - off1) == rd32(src + ip + 1)) {   /* :190-195 */
                mLength = zo_count(src, ip + 1 + 4, ip + 1 + 4 - off1, n) + 4;
                ip++;
                zo_store_seq(st, src, anchor, ip - anchor, 1, (uint32_t)mLength);
                kind = 1; break;
            }
            hl1 = zo_hash(src + ip1, hL, 8);
            if (idxl0 && rd64(src + idxl0 - 1) == rd64(src + ip)) {              /* :203-211 (idx >= prefixLowest) */
                size_t m = idxl0 - 1;
                mLength = zo_count(src, ip + 8, m + 8, n) + 8;
                offset = (uint32_t)(ip - m);
                while (ip > anchor && m > 0 && src[ip - 1] == src[m - 1]) { ip--; m--; mLength++; }
                kind = 2; break;
            }
            idxl1 = TL[hl1];
            if (idxs0 && rd32(src + idxs0 - 1) == rd32(src + ip)) {              /* :217-222 -> _search_next_long :253-271 */
                matchs0 = idxs0 - 1;
                mLength = zo_count(src, ip + 4, matchs0 + 4, n) + 4;
                offset = (uint32_t)(ip - matchs0);
                if (idxl1 > 1 && rd64(src + idxl1 - 1) == rd64(src + ip1)) {     /* :260 idxl1 > prefixLowestIndex (strict) */
                    size_t const m1 = idxl1 - 1;
                    size_t const l1len = zo_count(src, ip1 + 8, m1 + 8, n) + 8;
                    if (l1len > mLength) { ip = ip1; mLength = l1len; offset = (uint32_t)(ip - m1); matchs0 = m1; }
                }
                while (ip > anchor && matchs0 > 0 && src[ip - 1] == src[matchs0 - 1]) { ip--; matchs0--; mLength++; }
                kind = 2; break;
            }
            if (ip1 >= nextStep) { step++; nextStep += 256; }                    /* :224-229 */
            ip = ip1; ip1 += step;
            hl0 = hl1; idxl0 = idxl1;
        } while (ip1 <= ilimit);
        if (!kind) break;                                                        /* _cleanup */
        if (kind == 2) {                                                         /* _match_found :275-290 */
            off2 = off1; off1 = offset;
            if (step < 4) TL[hl1] = (uint32_t)ip1 + 1;
            zo_store_seq(st, src, anchor, ip - anchor, offset + 3, (uint32_t)mLength);
        }
        ip += mLength; anchor = ip;                                              /* _match_stored :292-321 */
        if (ip <= ilimit) {
            size_t const ins = curr + 2;
            TL[zo_hash(src + ins, hL, 8)] = (uint32_t)ins + 1;
            TL[zo_hash(src + ip - 2, hL, 8)] = (uint32_t)ip - 2 + 1;
            TS[zo_hash(src + ins, hS, mls)] = (uint32_t)ins + 1;
            TS[zo_hash(src + ip - 1, hS, mls)] = (uint32_t)ip - 1 + 1;
            while (ip <= ilimit && off2 > 0 && rd32(src + ip) == rd32(src + ip - off2)) {
                uint32_t const rLength = zo_count(src, ip + 4, ip + 4 - off2, n) + 4;
                uint32_t const t = off2; off2 = off1; off1 = t;
                TS[zo_hash(src + ip, hS, mls)] = (uint32_t)ip + 1;
                TL[zo_hash(src + ip, hL, 8)] = (uint32_t)ip + 1;
                zo_store_seq(st, src, anchor, 0, 1, rLength);
                ip += rLength; anchor = ip;
            }
        }
    }
    saved2 = (saved1 != 0 && off1 != 0) ? saved1 : saved2;                       /* :244-248 */
    rep[0] = off1 ? off1 : saved1;
    rep[1] = off2 ? off2 : saved2;
    free(TL); free(TS);
    return n - anchor;
}

/* ---- hash-chain match finder + lazy parser (strategies greedy / lazy / lazy2 with the row matcher disabled) ---- */
typedef struct {
    uint32_t* head;        /* hashTable: pos+1 of the most recent inserted position per hash, 0 = empty */
    uint32_t* chain;       /* chainTable[pos & mask]: pos+1 of the previous head at insertion time      */
    unsigned hlog, clog, slog, mls;
    size_t nextToUpdate;   /* zstd_compress_internal.h:232 */
    int lazySkipping;      /* :253 */
} zo_hc;

/* zstd_lazy.c:632-657 ZSTD_insertAndFindFirstIndex_internal + :667-773 ZSTD_HcFindBestMatch (noDict).
 * Returns the best match length (>= 4) at ip, or 3 when nothing was found; *offBase = offset + 3. */
static size_t zo_hc_best(zo_hc* hc, const uint8_t* src, size_t n, size_t ip, uint32_t* offBase)
{
    uint32_t const cmask = (1u << hc->clog) - 1;
    size_t const chainSize = (size_t)1 << hc->clog;
    unsigned nbAttempts = 1u << hc->slog;
    size_t ml = 4 - 1, idx = hc->nextToUpdate;
    uint32_t m;
    while (idx < ip) {                                                           /* :645-653 catch up */
        uint32_t const h = zo_hash(src + idx, hc->hlog, hc->mls);
        hc->chain[idx & cmask] = hc->head[h];
        hc->head[h] = (uint32_t)idx + 1;
        idx++;
        if (hc->lazySkipping) break;
    }
    hc->nextToUpdate = ip;
    m = hc->head[zo_hash(src + ip, hc->hlog, hc->mls)];
    /* lowLimit = start of the unit (:688-692: the window is never exceeded inside one <= 128 KB unit) */
    for (; m != 0 && nbAttempts > 0; nbAttempts--) {                             /* :711 */
        size_t const mp = m - 1;
        size_t cur = 0;
        if (rd32(src + mp + ml - 3) == rd32(src + ip + ml - 3)) cur = zo_count(src, ip, mp, n);   /* :716-717 */
        if (cur > ml) {                                                          /* :726-730 */
            ml = cur; *offBase = (uint32_t)(ip - mp) + 3;
            if (ip + cur == n) break;
        }
        if (ip >= chainSize && mp <= ip - chainSize) break;                      /* :732 matchIndex <= minChain */
        m = hc->chain[mp & cmask];
    }
    return ml;
}

/* ---- row-hash match finder (the reference's DEFAULT for greedy / lazy / lazy2 when windowLog > 14, zstd_compress.c:237-253) ----
 * zstd_lazy.c:778-960 (rows, tags, head rotation, update with the 384-position skip rule), :1141-1340 ZSTD_RowFindBestMatch (noDict).
 * The hash is salted (zstd_compress_internal.h:865-879); a FRESH CCtx starts from salt 0 / entropy 0 and advances the salt once
 * in ZSTD_reset_matchState before its first frame (zstd_compress.c:1964-1975, :2027-2033) — that constant is the salt here: the
 * parity target is ZSTD_compress2 on a fresh CCtx per unit (a reused CCtx mixes the previous frames' hashes into the salt). */
static int g_zo_row_matcher = 1;              /* 1 = reference default (auto), 0 = ZSTD_c_useRowMatchFinder = ZSTD_ps_disable */
void zo_set_row_matcher(int enable) { g_zo_row_matcher = enable; }
static uint64_t zo_rotr64(uint64_t v, unsigned c) { return (v >> c) | (v << (64 - c)); }
static uint64_t zo_bitmix(uint64_t val, uint64_t len)                            /* zstd_compress.c:1964-1970 */
{
    val ^= zo_rotr64(val, 49) ^ zo_rotr64(val, 24);
    val *= 0x9FB21C651E98DF25ULL;
    val ^= (val >> 35) + len;
    val *= 0x9FB21C651E98DF25ULL;
    return val ^ (val >> 28);
}
unsigned long long zo_fresh_hash_salt(void) { return zo_bitmix(0, 8) ^ zo_bitmix(0, 4); }   /* ZSTD_advanceHashSalt from (0, 0) */
static uint32_t zo_hash_salted(const uint8_t* p, unsigned hBits, unsigned mls, uint64_t salt)   /* internal.h:820-879 */
{
    switch (mls) {
    default:
    case 4: return ((rd32(p) * 2654435761U) ^ (uint32_t)salt) >> (32 - hBits);
    case 5: return (uint32_t)((((rd64(p) << 24) * 889523592379ULL) ^ salt) >> (64 - hBits));
    case 6: return (uint32_t)((((rd64(p) << 16) * 227718039650203ULL) ^ salt) >> (64 - hBits));
    }
}
typedef struct {
    uint32_t* row;         /* hashTable: rows of 1 << rowLog entries, pos+1, 0 = empty */
    uint8_t* tag;          /* tagTable: byte 0 of every row is its head */
    unsigned rowHashLog, rowLog, slog, mls;
    uint64_t salt;
    size_t nextToUpdate;
    int lazySkipping;
} zo_row;
static unsigned zo_row_next_index(uint8_t* tagRow, unsigned rowMask)              /* zstd_lazy.c:797-803 */
{
    unsigned next = ((unsigned)tagRow[0] - 1) & rowMask;
    next += (next == 0) ? rowMask : 0;
    tagRow[0] = (uint8_t)next;
    return next;
}
static void zo_row_insert_range(zo_row* r, const uint8_t* src, size_t from, size_t to)   /* :880-910 */
{
    unsigned const rowMask = (1u << r->rowLog) - 1;
    for (; from < to; from++) {
        uint32_t const h = zo_hash_salted(src + from, r->rowHashLog + 8, r->mls, r->salt);
        size_t const rel = (size_t)(h >> 8) << r->rowLog;
        unsigned const pos = zo_row_next_index(r->tag + rel, rowMask);
        r->tag[rel + pos] = (uint8_t)h;
        r->row[rel + pos] = (uint32_t)from + 1;
    }
}
/* analysis hook ("two-pass prediction"): log the ranges the 384-position rule leaves out; in PREDICT mode the rule is
 * only logged, not applied — the parse a device pass would make from records that assume every position inserted */
static int g_zo_row_predict = 0;
static size_t g_zo_skiplog[2 * 4096]; static size_t g_zo_skiplog_n = 0;
void zo_row_analysis(int predict) { g_zo_row_predict = predict; g_zo_skiplog_n = 0; }
size_t zo_row_skiplog(size_t* out, size_t cap) { size_t i, n = g_zo_skiplog_n < cap ? g_zo_skiplog_n : cap; for (i = 0; i < 2 * n; i++) out[i] = g_zo_skiplog[i]; return g_zo_skiplog_n; }
static void zo_row_update(zo_row* r, const uint8_t* src, size_t target)          /* :916-947 (useCache) */
{
    size_t idx = r->nextToUpdate;
    if (target - idx > 384) {                                                    /* kSkipThreshold: only the first 96 and the last 32 */
        if (g_zo_skiplog_n < 4096) { g_zo_skiplog[2 * g_zo_skiplog_n] = idx + 96; g_zo_skiplog[2 * g_zo_skiplog_n + 1] = target - 32; g_zo_skiplog_n++; }
        if (g_zo_row_predict) { zo_row_insert_range(r, src, idx, target); r->nextToUpdate = target; return; }
        zo_row_insert_range(r, src, idx, idx + 96);
        idx = target - 32;
    }
    zo_row_insert_range(r, src, idx, target);
    r->nextToUpdate = target;
}
static size_t zo_row_best(zo_row* r, const uint8_t* src, size_t n, size_t ip, uint32_t* offBase)   /* :1141-1340 */
{
    unsigned const rowEntries = 1u << r->rowLog, rowMask = rowEntries - 1;
    unsigned const capped = r->slog < r->rowLog ? r->slog : r->rowLog;
    unsigned nbAttempts = 1u << capped, numMatches = 0, k;
    uint32_t buf[64];
    size_t ml = 4 - 1;
    uint32_t h;
    if (!r->lazySkipping) zo_row_update(r, src, ip);
    else r->nextToUpdate = ip;
    h = zo_hash_salted(src + ip, r->rowHashLog + 8, r->mls, r->salt);
    {   size_t const rel = (size_t)(h >> 8) << r->rowLog;
        uint8_t* const tagRow = r->tag + rel; uint32_t* const row = r->row + rel;
        unsigned const head = tagRow[0] & rowMask;
        for (k = 0; k < rowEntries && nbAttempts > 0; k++) {                      /* :1232-1247 entries from the most recent on */
            unsigned const pos = (head + k) & rowMask;
            if (tagRow[pos] != (uint8_t)h) continue;
            if (pos == 0) continue;
            if (row[pos] == 0) break;                                            /* matchIndex < lowLimit: an empty slot */
            buf[numMatches++] = row[pos] - 1;
            nbAttempts--;
        }
        {   unsigned const pos = zo_row_next_index(tagRow, rowMask);             /* :1251-1255 the searched position goes in too */
            tagRow[pos] = (uint8_t)h;
            row[pos] = (uint32_t)r->nextToUpdate + 1;
            r->nextToUpdate++;
        }
    }
    for (k = 0; k < numMatches; k++) {                                            /* :1258-1285 */
        size_t const mp = buf[k];
        size_t cur = 0;
        if (rd32(src + mp + ml - 3) == rd32(src + ip + ml - 3)) cur = zo_count(src, ip, mp, n);
        if (cur > ml) { ml = cur; *offBase = (uint32_t)(ip - mp) + 3; if (ip + cur == n) break; }
    }
    return ml;
}

static unsigned zo_gain_bits(uint32_t offBase) { return hb32(offBase); }

/* zstd_lazy.c:1516-1779 ZSTD_compressBlock_lazy_generic(search_hashChain, depth, ZSTD_noDict), one block, empty history */
static size_t zo_lazy(const zo_cparams* cp, const uint8_t* src, size_t n, zo_store* st, uint32_t rep[3], unsigned depth)
{
    zo_hc hc; zo_row rw;
    int const useRow = g_zo_row_matcher && cp->windowLog > 14;                   /* zstd_compress.c:237-253 ZSTD_resolveRowMatchFinderMode */
    if (useRow && n < 16) return n;                                              /* ilimit = n - 16 < 0: nothing is searched */
    size_t const ilimit = useRow ? n - 16 : n - 8;                               /* :1527 the row matcher stops ZSTD_ROW_HASH_CACHE_SIZE earlier */
    size_t ip = 0, anchor = 0;
    uint32_t off1 = rep[0], off2 = rep[1], saved1 = 0, saved2 = 0;
    hc.hlog = cp->hashLog; hc.clog = cp->chainLog; hc.slog = cp->searchLog;
    hc.mls = cp->minMatch < 4 ? 4 : (cp->minMatch > 6 ? 6 : cp->minMatch);      /* :1531 */
    hc.head = (uint32_t*)calloc((size_t)1 << hc.hlog, sizeof(uint32_t));
    hc.chain = (uint32_t*)calloc((size_t)1 << hc.clog, sizeof(uint32_t));
    hc.nextToUpdate = 0; hc.lazySkipping = 0;                                    /* :1567 */
    rw.rowLog = cp->searchLog < 4 ? 4 : (cp->searchLog > 6 ? 6 : cp->searchLog); /* zstd_compress.c:2042 */
    rw.rowHashLog = cp->hashLog - rw.rowLog; rw.slog = cp->searchLog; rw.mls = hc.mls;
    rw.row = (uint32_t*)calloc((size_t)1 << cp->hashLog, sizeof(uint32_t));
    rw.tag = (uint8_t*)calloc((size_t)1 << cp->hashLog, 1);
    rw.salt = zo_fresh_hash_salt(); rw.nextToUpdate = 0; rw.lazySkipping = 0;
#define ZO_BEST(ipx, ob) (useRow ? zo_row_best(&rw, src, n, (ipx), (ob)) : zo_hc_best(&hc, src, n, (ipx), (ob)))
    ip += 1;                                                                     /* :1552 dictAndPrefixLength == 0 */
    {   uint32_t const maxRep = 1;                                               /* :1553-1559 */
        if (off2 > maxRep) { saved2 = off2; off2 = 0; }
        if (off1 > maxRep) { saved1 = off1; off1 = 0; }
    }
    while (ip < ilimit) {                                                        /* :1581 */
        size_t matchLength = 0, start = ip + 1;
        uint32_t offBase = 1;                                                    /* REPCODE1_TO_OFFBASE */
        int direct = 0;
        if (off1 > 0 && rd32(src + ip + 1 - off1) == rd32(src + ip + 1)) {       /* :1600-1604 */
            matchLength = zo_count(src, ip + 1 + 4, ip + 1 + 4 - off1, n) + 4;
            if (depth == 0) direct = 1;
        }
        if (!direct) {
            {   uint32_t found = 999999999;                                      /* :1607-1611 */
                size_t const ml2 = ZO_BEST(ip, &found);
                if (ml2 > matchLength) { matchLength = ml2; start = ip; offBase = found; }
            }
            if (matchLength < 4) {                                               /* :1613-1625 */
                size_t const step = ((ip - anchor) >> 8) + 1;                    /* kSearchStrength = 8 */
                ip += step;
                hc.lazySkipping = rw.lazySkipping = step > 8;                    /* kLazySkippingStep = 8 */
                continue;
            }
            if (depth >= 1)
            while (ip < ilimit) {                                                /* :1628-1700 */
                ip++;
                if (offBase && off1 > 0 && rd32(src + ip) == rd32(src + ip - off1)) {
                    size_t const mlRep = zo_count(src, ip + 4, ip + 4 - off1, n) + 4;
                    int const gain2 = (int)(mlRep * 3);
                    int const gain1 = (int)(matchLength * 3 - zo_gain_bits(offBase) + 1);
                    if (mlRep >= 4 && gain2 > gain1) { matchLength = mlRep; offBase = 1; start = ip; }
                }
                {   uint32_t cand = 999999999;
                    size_t const ml2 = ZO_BEST(ip, &cand);
                    int const gain2 = (int)(ml2 * 4 - zo_gain_bits(cand));
                    int const gain1 = (int)(matchLength * 4 - zo_gain_bits(offBase) + 4);
                    if (ml2 >= 4 && gain2 > gain1) { matchLength = ml2; offBase = cand; start = ip; continue; }
                }
                if (depth == 2 && ip < ilimit) {                                 /* :1663-1698 */
                    ip++;
                    if (offBase && off1 > 0 && rd32(src + ip) == rd32(src + ip - off1)) {
                        size_t const mlRep = zo_count(src, ip + 4, ip + 4 - off1, n) + 4;
                        int const gain2 = (int)(mlRep * 4);
                        int const gain1 = (int)(matchLength * 4 - zo_gain_bits(offBase) + 1);
                        if (mlRep >= 4 && gain2 > gain1) { matchLength = mlRep; offBase = 1; start = ip; }
                    }
                    {   uint32_t cand = 999999999;
                        size_t const ml2 = ZO_BEST(ip, &cand);
                        int const gain2 = (int)(ml2 * 4 - zo_gain_bits(cand));
                        int const gain1 = (int)(matchLength * 4 - zo_gain_bits(offBase) + 7);
                        if (ml2 >= 4 && gain2 > gain1) { matchLength = ml2; offBase = cand; start = ip; continue; }
                    }
                }
                break;
            }
            if (offBase > 3) {                                                   /* :1707-1714 catch up */
                uint32_t const off = offBase - 3;
                while (start > anchor && start - off > 0 && src[start - 1] == src[start - off - 1]) { start--; matchLength++; }
                off2 = off1; off1 = off;
            }
        }
        zo_store_seq(st, src, anchor, start - anchor, offBase, (uint32_t)matchLength);   /* :1727-1731 */
        anchor = ip = start + matchLength;
        hc.lazySkipping = rw.lazySkipping = 0;                                   /* :1732-1738 */
        while (ip <= ilimit && off2 > 0 && rd32(src + ip) == rd32(src + ip - off2)) {    /* :1763-1773 */
            uint32_t const t = off2;
            matchLength = zo_count(src, ip + 4, ip + 4 - off2, n) + 4;
            off2 = off1; off1 = t;
            zo_store_seq(st, src, anchor, 0, 1, (uint32_t)matchLength);
            ip += matchLength; anchor = ip;
        }
    }
    saved2 = (saved1 != 0 && off1 != 0) ? saved1 : saved2;                       /* :1777-1783 */
    rep[0] = off1 ? off1 : saved1;
    rep[1] = off2 ? off2 : saved2;
    free(hc.head); free(hc.chain); free(rw.row); free(rw.tag);
#undef ZO_BEST
    return n - anchor;
}

/* ---- the same parser for the blocks of a multi-block frame: the match state lives across blocks (zstd_lazy.c:1516-1779 with a
 * prefix; positions are relative to the frame start, the reference's indices are position + 2).  S->low = window.lowLimit =
 * window.dictLimit as a position: what ZSTD_window_enforceMaxDist (zstd_compress.c:4555, called with the block START) has left. */
typedef struct { zo_hc hc; zo_row rw; int useRow; size_t low; unsigned windowLog; } zo_lz;

static size_t zo_lz_low_limit(const zo_lz* S, size_t ip)                          /* ZSTD_getLowestMatchIndex, no dictionary (internal.h:1312) */
{
    size_t const maxDist = (size_t)1 << S->windowLog;
    return (ip - S->low > maxDist) ? ip - maxDist : S->low;
}
static size_t zo_hc_best_w(zo_lz* S, const uint8_t* src, size_t bEnd, size_t ip, uint32_t* offBase)   /* :667-773 */
{
    zo_hc* const hc = &S->hc;
    uint32_t const cmask = (1u << hc->clog) - 1;
    size_t const chainSize = (size_t)1 << hc->clog;
    size_t const lowLimit = zo_lz_low_limit(S, ip);
    unsigned nbAttempts = 1u << hc->slog;
    size_t ml = 4 - 1, idx = hc->nextToUpdate;
    uint32_t m;
    while (idx < ip) {                                                           /* :645-653 */
        uint32_t const h = zo_hash(src + idx, hc->hlog, hc->mls);
        hc->chain[idx & cmask] = hc->head[h];
        hc->head[h] = (uint32_t)idx + 1;
        idx++;
        if (hc->lazySkipping) break;
    }
    hc->nextToUpdate = ip;
    m = hc->head[zo_hash(src + ip, hc->hlog, hc->mls)];
    for (; m != 0 && (size_t)(m - 1) >= lowLimit && nbAttempts > 0; nbAttempts--) {   /* :711 matchIndex >= lowLimit */
        size_t const mp = m - 1;
        size_t cur = 0;
        if (rd32(src + mp + ml - 3) == rd32(src + ip + ml - 3)) cur = zo_count(src, ip, mp, bEnd);
        if (cur > ml) {
            ml = cur; *offBase = (uint32_t)(ip - mp) + 3;
            if (ip + cur == bEnd) break;
        }
        if (ip + 2 > chainSize && mp + chainSize <= ip) break;                   /* :732 matchIndex <= minChain, indices = position + 2 */
        m = hc->chain[mp & cmask];
    }
    return ml;
}
static size_t zo_row_best_w(zo_lz* S, const uint8_t* src, size_t bEnd, size_t ip, uint32_t* offBase)   /* :1141-1340 */
{
    zo_row* const r = &S->rw;
    unsigned const rowEntries = 1u << r->rowLog, rowMask = rowEntries - 1;
    unsigned const capped = r->slog < r->rowLog ? r->slog : r->rowLog;
    size_t const lowLimit = zo_lz_low_limit(S, ip);
    unsigned nbAttempts = 1u << capped, numMatches = 0, k;
    uint32_t buf[64];
    size_t ml = 4 - 1;
    uint32_t h;
    if (!r->lazySkipping) zo_row_update(r, src, ip);
    else r->nextToUpdate = ip;
    h = zo_hash_salted(src + ip, r->rowHashLog + 8, r->mls, r->salt);
    {   size_t const rel = (size_t)(h >> 8) << r->rowLog;
        uint8_t* const tagRow = r->tag + rel; uint32_t* const row = r->row + rel;
        unsigned const head = tagRow[0] & rowMask;
        for (k = 0; k < rowEntries && nbAttempts > 0; k++) {
            unsigned const pos = (head + k) & rowMask;
            if (tagRow[pos] != (uint8_t)h) continue;
            if (pos == 0) continue;
            if (row[pos] == 0 || (size_t)(row[pos] - 1) < lowLimit) break;       /* :1235 matchIndex < lowLimit */
            buf[numMatches++] = row[pos] - 1;
            nbAttempts--;
        }
        {   unsigned const pos = zo_row_next_index(tagRow, rowMask);
            tagRow[pos] = (uint8_t)h;
            row[pos] = (uint32_t)r->nextToUpdate + 1;
            r->nextToUpdate++;
        }
    }
    for (k = 0; k < numMatches; k++) {
        size_t const mp = buf[k];
        size_t cur = 0;
        if (rd32(src + mp + ml - 3) == rd32(src + ip + ml - 3)) cur = zo_count(src, ip, mp, bEnd);
        if (cur > ml) { ml = cur; *offBase = (uint32_t)(ip - mp) + 3; if (ip + cur == bEnd) break; }
    }
    return ml;
}
static int zo_lz_init(zo_lz* S, const zo_cparams* cp)
{
    memset(S, 0, sizeof(*S));
    S->useRow = g_zo_row_matcher && cp->windowLog > 14; S->windowLog = cp->windowLog;
    S->hc.hlog = cp->hashLog; S->hc.clog = cp->chainLog; S->hc.slog = cp->searchLog;
    S->hc.mls = cp->minMatch < 4 ? 4 : (cp->minMatch > 6 ? 6 : cp->minMatch);
    S->rw.rowLog = cp->searchLog < 4 ? 4 : (cp->searchLog > 6 ? 6 : cp->searchLog);
    S->rw.rowHashLog = cp->hashLog - S->rw.rowLog; S->rw.slog = cp->searchLog; S->rw.mls = S->hc.mls;
    S->rw.salt = zo_fresh_hash_salt();
    if (S->useRow) { S->rw.row = (uint32_t*)calloc((size_t)1 << cp->hashLog, sizeof(uint32_t)); S->rw.tag = (uint8_t*)calloc((size_t)1 << cp->hashLog, 1); return S->rw.row && S->rw.tag; }
    S->hc.head = (uint32_t*)calloc((size_t)1 << S->hc.hlog, sizeof(uint32_t));
    S->hc.chain = (uint32_t*)calloc((size_t)1 << S->hc.clog, sizeof(uint32_t));
    return S->hc.head && S->hc.chain;
}
static void zo_lz_free(zo_lz* S) { free(S->hc.head); free(S->hc.chain); free(S->rw.row); free(S->rw.tag); }

/* one block [bStart, bStart + bLen) of the frame; returns the trailing literals */
static size_t zo_lazy_block(const zo_cparams* cp, const uint8_t* src, size_t bStart, size_t bLen, zo_lz* S, zo_store* st, uint32_t rep[3], unsigned depth)
{
    size_t const maxDist = (size_t)1 << cp->windowLog;
    size_t const bEnd = bStart + bLen;
    size_t const guard = S->useRow ? 16 : 8;                                     /* :1527 */
    size_t ilimit, ip = bStart, anchor = bStart;
    uint32_t off1 = rep[0], off2 = rep[1], saved1 = 0, saved2 = 0;
    size_t* const ntu = S->useRow ? &S->rw.nextToUpdate : &S->hc.nextToUpdate;
    /* zstd_compress.c:4555 ZSTD_window_enforceMaxDist(block start), then :3243-3249 "limited update after a very long match" */
    if (bStart + 2 > maxDist && bStart > maxDist && bStart - maxDist > S->low) S->low = bStart - maxDist;
    if (bStart > *ntu + 384) { size_t const d = bStart - *ntu - 384; *ntu = bStart - (d < 192 ? d : 192); }
    S->hc.lazySkipping = S->rw.lazySkipping = 0;                                 /* :1567 */
    if (bLen < guard) return bLen;
    ilimit = bEnd - guard;
#define ZO_BESTW(ipx, ob) (S->useRow ? zo_row_best_w(S, src, bEnd, (ipx), (ob)) : zo_hc_best_w(S, src, bEnd, (ipx), (ob)))
    ip += (ip == S->low);                                                        /* :1552 dictAndPrefixLength == 0 */
    {   size_t const windowLow = (ip - S->low > maxDist) ? ip - maxDist : S->low;   /* :1553-1559 ZSTD_getLowestPrefixIndex */
        size_t const maxRep = ip - windowLow;
        if (off2 > maxRep) { saved2 = off2; off2 = 0; }
        if (off1 > maxRep) { saved1 = off1; off1 = 0; }
    }
    while (ip < ilimit) {
        size_t matchLength = 0, start = ip + 1;
        uint32_t offBase = 1;
        int direct = 0;
        if (off1 > 0 && rd32(src + ip + 1 - off1) == rd32(src + ip + 1)) {
            matchLength = zo_count(src, ip + 1 + 4, ip + 1 + 4 - off1, bEnd) + 4;
            if (depth == 0) direct = 1;
        }
        if (!direct) {
            {   uint32_t found = 999999999;
                size_t const ml2 = ZO_BESTW(ip, &found);
                if (ml2 > matchLength) { matchLength = ml2; start = ip; offBase = found; }
            }
            if (matchLength < 4) {
                size_t const step = ((ip - anchor) >> 8) + 1;
                ip += step;
                S->hc.lazySkipping = S->rw.lazySkipping = step > 8;
                continue;
            }
            if (depth >= 1)
            while (ip < ilimit) {
                ip++;
                if (offBase && off1 > 0 && rd32(src + ip) == rd32(src + ip - off1)) {
                    size_t const mlRep = zo_count(src, ip + 4, ip + 4 - off1, bEnd) + 4;
                    int const gain2 = (int)(mlRep * 3);
                    int const gain1 = (int)(matchLength * 3 - zo_gain_bits(offBase) + 1);
                    if (mlRep >= 4 && gain2 > gain1) { matchLength = mlRep; offBase = 1; start = ip; }
                }
                {   uint32_t cand = 999999999;
                    size_t const ml2 = ZO_BESTW(ip, &cand);
                    int const gain2 = (int)(ml2 * 4 - zo_gain_bits(cand));
                    int const gain1 = (int)(matchLength * 4 - zo_gain_bits(offBase) + 4);
                    if (ml2 >= 4 && gain2 > gain1) { matchLength = ml2; offBase = cand; start = ip; continue; }
                }
                if (depth == 2 && ip < ilimit) {
                    ip++;
                    if (offBase && off1 > 0 && rd32(src + ip) == rd32(src + ip - off1)) {
                        size_t const mlRep = zo_count(src, ip + 4, ip + 4 - off1, bEnd) + 4;
                        int const gain2 = (int)(mlRep * 4);
                        int const gain1 = (int)(matchLength * 4 - zo_gain_bits(offBase) + 1);
                        if (mlRep >= 4 && gain2 > gain1) { matchLength = mlRep; offBase = 1; start = ip; }
                    }
                    {   uint32_t cand = 999999999;
                        size_t const ml2 = ZO_BESTW(ip, &cand);
                        int const gain2 = (int)(ml2 * 4 - zo_gain_bits(cand));
                        int const gain1 = (int)(matchLength * 4 - zo_gain_bits(offBase) + 7);
                        if (ml2 >= 4 && gain2 > gain1) { matchLength = ml2; offBase = cand; start = ip; continue; }
                    }
                }
                break;
            }
            if (offBase > 3) {                                                   /* :1707-1714: match start > prefixLowest */
                uint32_t const off = offBase - 3;
                while (start > anchor && start - off > S->low && src[start - 1] == src[start - off - 1]) { start--; matchLength++; }
                off2 = off1; off1 = off;
            }
        }
        zo_store_seq(st, src, anchor, start - anchor, offBase, (uint32_t)matchLength);
        anchor = ip = start + matchLength;
        S->hc.lazySkipping = S->rw.lazySkipping = 0;
        while (ip <= ilimit && off2 > 0 && rd32(src + ip) == rd32(src + ip - off2)) {
            uint32_t const t = off2;
            matchLength = zo_count(src, ip + 4, ip + 4 - off2, bEnd) + 4;
            off2 = off1; off1 = t;
            zo_store_seq(st, src, anchor, 0, 1, (uint32_t)matchLength);
            ip += matchLength; anchor = ip;
        }
    }
    saved2 = (saved1 != 0 && off1 != 0) ? saved1 : saved2;
    rep[0] = off1 ? off1 : saved1;
    rep[1] = off2 ? off2 : saved2;
#undef ZO_BESTW
    return bEnd - anchor;
}

/* zstd_compress.c:3207-3369 ZSTD_buildSeqStore for a history-less block (+ :3365 trailing literals) */
size_t zo_parse_block(const zo_cparams* cp, const uint8_t* src, size_t n,
                      zo_seq* seqs, size_t cap, uint8_t* lits, size_t* litSize, uint32_t repOut[3])
{
    zo_store st; uint32_t rep[3] = {1, 4, 8};                                    /* zstd_internal.h:69 */
    size_t last;
    st.seqs = seqs; st.nb = 0; st.cap = cap; st.lits = lits; st.litSize = 0; st.overflow = 0;
    if (n < 8) last = n;                     /* the block compressors' loops need ilimit = n-8 >= 0; nothing found */
    else if (cp->strategy == 1) last = zo_fast(cp, src, n, &st, rep);
    else if (cp->strategy == 2) last = zo_dfast(cp, src, n, &st, rep);
    else last = zo_lazy(cp, src, n, &st, rep, cp->strategy - 3);                 /* greedy 3 / lazy 4 / lazy2 5 */
    memcpy(lits + st.litSize, src + n - last, last);
    st.litSize += last;
    *litSize = st.litSize;
    if (repOut) { repOut[0] = rep[0]; repOut[1] = rep[1]; repOut[2] = rep[2]; }
    return st.overflow ? ZO_ERROR : st.nb;
}

/* zstd_compress.c:3371-3454 ZSTD_copyBlockSequences; repcode resolution per zstd_compress_internal.h:735 */
size_t zo_sequences_public(const zo_cparams* cp, const uint8_t* src, size_t n, uint32_t* out, size_t capSeqs)
{
    zo_seq* seqs = (zo_seq*)malloc(sizeof(zo_seq) * (n / 3 + 2));
    uint8_t* lits = (uint8_t*)malloc(n + 8);
    size_t litSize = 0, i, sumLit = 0;
    uint32_t rep[3] = {1, 4, 8};
    size_t const nb = zo_parse_block(cp, src, n, seqs, n / 3 + 2, lits, &litSize, NULL);
    if (nb == ZO_ERROR || nb + 1 > capSeqs) { free(seqs); free(lits); return ZO_ERROR; }
    for (i = 0; i < nb; i++) {
        uint32_t const ob = seqs[i].offBase, ll = seqs[i].litLength;
        uint32_t raw, repField = 0;
        if (ob <= 3) {
            repField = ob;
            if (ll != 0) raw = rep[ob - 1];
            else raw = (ob == 3) ? rep[0] - 1 : rep[ob];
        } else raw = ob - 3;
        out[4*i] = raw; out[4*i+1] = ll; out[4*i+2] = seqs[i].matchLength; out[4*i+3] = repField;
        if (ob > 3) { rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = ob - 3; }
        else {
            uint32_t const rc = ob - 1 + (ll == 0);
            if (rc > 0) {
                uint32_t const cur = (rc == 3) ? rep[0] - 1 : rep[rc];
                rep[2] = (rc >= 2) ? rep[1] : rep[2];
                rep[1] = rep[0]; rep[0] = cur;
            }
        }
        sumLit += ll;
    }
    out[4*nb] = 0; out[4*nb+1] = (uint32_t)(litSize - sumLit); out[4*nb+2] = 0; out[4*nb+3] = 0;
    free(seqs); free(lits);
    return nb + 1;
}

/* ------------------------------------------------------------------ bit writers */
/* lib/common/bitstream.h:145-231: LSB-first stream, closed by a single 1 bit. */
typedef struct { uint8_t* p; uint64_t acc; unsigned nb; } zo_bits;
static void bw_init(zo_bits* b, uint8_t* dst) { b->p = dst; b->acc = 0; b->nb = 0; }
static void bw_add(zo_bits* b, uint64_t v, unsigned n)
{
    if (n == 0) return;
    b->acc |= (v & ((1ULL << n) - 1)) << b->nb;
    b->nb += n;
    while (b->nb >= 8) { *b->p++ = (uint8_t)b->acc; b->acc >>= 8; b->nb -= 8; }
}
static uint8_t* bw_close(zo_bits* b)            /* bitstream.h:222: add 1 bit, flush, +1 byte if bits pending */
{
    bw_add(b, 1, 1);
    if (b->nb) { *b->p++ = (uint8_t)b->acc; b->acc = 0; b->nb = 0; }
    return b->p;
}

/* ------------------------------------------------------------------ histogram */
size_t zo_hist(unsigned count[256], unsigned* maxSym, const uint8_t* src, size_t n)   /* hist.c:29-59 */
{
    size_t i; unsigned largest = 0, m = 255, s;
    memset(count, 0, 256 * sizeof(unsigned));
    if (n == 0) { *maxSym = 0; return 0; }
    for (i = 0; i < n; i++) count[src[i]]++;
    while (!count[m]) m--;
    for (s = 0; s <= m; s++) if (count[s] > largest) largest = count[s];
    *maxSym = m;
    return largest;
}
static size_t hist_small(unsigned* count, unsigned* maxSym, const uint8_t* src, size_t n)  /* same, alphabet <= *maxSym */
{
    size_t i; unsigned largest = 0, m = *maxSym, s;
    memset(count, 0, (m + 1) * sizeof(unsigned));
    if (n == 0) { *maxSym = 0; return 0; }
    for (i = 0; i < n; i++) count[src[i]]++;
    while (!count[m]) m--;
    for (s = 0; s <= m; s++) if (count[s] > largest) largest = count[s];
    *maxSym = m;
    return largest;
}

/* ------------------------------------------------------------------ FSE */
typedef struct {
    unsigned tableLog;
    uint16_t state[512];            /* next-state table, sorted by symbol (fse_compress.c:170-173) */
    int32_t  dFind[64];             /* deltaFindState */
    uint32_t dBits[64];             /* deltaNbBits    */
    unsigned maxSym;                /* FSE_CTable header: maxSymbolValue (what ZSTD_getFSEMaxSymbolValue reads) */
} zo_fse;

/* fse_compress.c:348-374 */
static unsigned fse_min_log(size_t n, unsigned maxSym)
{
    unsigned const a = hb32((uint32_t)n) + 1, b = hb32(maxSym) + 2;
    return a < b ? a : b;
}
static unsigned fse_optimal_log(unsigned maxLog, size_t n, unsigned maxSym, unsigned minus)
{
    unsigned const maxBitsSrc = hb32((uint32_t)(n - 1)) - minus;
    unsigned log = maxLog, minBits = fse_min_log(n, maxSym);
    if (maxBitsSrc < log) log = maxBitsSrc;
    if (minBits > log) log = minBits;
    if (log < 5) log = 5;
    if (log > 12) log = 12;
    return log;
}

/* fse_compress.c:379-463 */
static int fse_normalize_m2(short* norm, unsigned tableLog, const unsigned* count, size_t total, unsigned maxSym, short lowProb)
{
    unsigned s, distributed = 0, toDistribute;
    uint32_t const lowThreshold = (uint32_t)(total >> tableLog);
    uint32_t lowOne = (uint32_t)((total * 3) >> (tableLog + 1));
    for (s = 0; s <= maxSym; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = lowProb; distributed++; total -= count[s]; continue; }
        if (count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; continue; }
        norm[s] = -2;
    }
    toDistribute = (1u << tableLog) - distributed;
    if (toDistribute == 0) return 0;
    if ((total / toDistribute) > lowOne) {
        lowOne = (uint32_t)((total * 3) / (toDistribute * 2));
        for (s = 0; s <= maxSym; s++)
            if (norm[s] == -2 && count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; }
        toDistribute = (1u << tableLog) - distributed;
    }
    if (distributed == maxSym + 1) {
        unsigned maxV = 0, maxC = 0;
        for (s = 0; s <= maxSym; s++) if (count[s] > maxC) { maxV = s; maxC = count[s]; }
        norm[maxV] += (short)toDistribute;
        return 0;
    }
    if (total == 0) {
        for (s = 0; toDistribute > 0; s = (s + 1) % (maxSym + 1))
            if (norm[s] > 0) { toDistribute--; norm[s]++; }
        return 0;
    }
    {   uint64_t const vStepLog = 62 - tableLog;
        uint64_t const mid = (1ULL << (vStepLog - 1)) - 1;
        uint64_t const rStep = ((((uint64_t)1 << vStepLog) * toDistribute) + mid) / (uint32_t)total;
        uint64_t tmpTotal = mid;
        for (s = 0; s <= maxSym; s++) {
            if (norm[s] == -2) {
                uint64_t const end = tmpTotal + (count[s] * rStep);
                uint32_t const sStart = (uint32_t)(tmpTotal >> vStepLog), sEnd = (uint32_t)(end >> vStepLog);
                if (sEnd - sStart < 1) return -1;
                norm[s] = (short)(sEnd - sStart);
                tmpTotal = end;
    }   }   }
    return 0;
}

/* fse_compress.c:465-525; returns tableLog, 0 for the rle special case, -1 on error */
int zo_fse_normalize(short* norm, unsigned tableLog, const unsigned* count, size_t total, unsigned maxSym, unsigned useLowProb)
{
    static const uint32_t rtb[8] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };
    short const lowProb = useLowProb ? -1 : 1;
    uint64_t const scale = 62 - tableLog;
    uint64_t const step = ((uint64_t)1 << 62) / (uint32_t)total;
    uint64_t const vStep = 1ULL << (scale - 20);
    int still = 1 << tableLog;
    unsigned s, largest = 0; short largestP = 0;
    uint32_t const lowThreshold = (uint32_t)(total >> tableLog);
    if (tableLog < 5 || tableLog > 12) return -1;
    if (tableLog < fse_min_log(total, maxSym)) return -1;
    for (s = 0; s <= maxSym; s++) {
        if (count[s] == total) return 0;
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = lowProb; still--; }
        else {
            short proba = (short)((count[s] * step) >> scale);
            if (proba < 8) {
                uint64_t const restToBeat = vStep * rtb[proba];
                proba += (count[s] * step) - ((uint64_t)proba << scale) > restToBeat;
            }
            if (proba > largestP) { largestP = proba; largest = s; }
            norm[s] = proba; still -= proba;
        }
    }
    if (-still >= (norm[largest] >> 1)) {
        if (fse_normalize_m2(norm, tableLog, count, total, maxSym, lowProb) < 0) return -1;
    } else norm[largest] += (short)still;
    return (int)tableLog;
}

/* fse_compress.c:234-327 (writeIsSafe path; callers provide room). returns size or 0 on error */
static size_t fse_write_ncount(uint8_t* out0, const short* norm, unsigned maxSym, unsigned tableLog)
{
    uint8_t* out = out0;
    int const tableSize = 1 << tableLog;
    int nbBits = (int)tableLog + 1, remaining = tableSize + 1, threshold = tableSize, bitCount = 4, previousIs0 = 0;
    uint32_t bitStream = tableLog - 5;
    unsigned symbol = 0; unsigned const alphabetSize = maxSym + 1;
    while (symbol < alphabetSize && remaining > 1) {
        if (previousIs0) {
            unsigned start = symbol;
            while (symbol < alphabetSize && !norm[symbol]) symbol++;
            if (symbol == alphabetSize) break;
            while (symbol >= start + 24) {
                start += 24; bitStream += 0xFFFFU << bitCount;
                out[0] = (uint8_t)bitStream; out[1] = (uint8_t)(bitStream >> 8); out += 2; bitStream >>= 16;
            }
            while (symbol >= start + 3) { start += 3; bitStream += 3U << bitCount; bitCount += 2; }
            bitStream += (symbol - start) << bitCount; bitCount += 2;
            if (bitCount > 16) {
                out[0] = (uint8_t)bitStream; out[1] = (uint8_t)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16;
            }
        }
        {   int count = norm[symbol++];
            int const max = (2 * threshold - 1) - remaining;
            remaining -= count < 0 ? -count : count;
            count++;
            if (count >= threshold) count += max;
            bitStream += (uint32_t)count << bitCount;
            bitCount += nbBits; bitCount -= (count < max);
            previousIs0 = (count == 1);
            if (remaining < 1) return 0;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
        if (bitCount > 16) {
            out[0] = (uint8_t)bitStream; out[1] = (uint8_t)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16;
        }
    }
    if (remaining != 1) return 0;
    out[0] = (uint8_t)bitStream; out[1] = (uint8_t)(bitStream >> 8);
    out += (bitCount + 7) / 8;
    return (size_t)(out - out0);
}

/* fse_compress.c:68-214 */
static void fse_build(zo_fse* ct, const short* norm, unsigned maxSym, unsigned tableLog)
{
    uint32_t const tableSize = 1u << tableLog, mask = tableSize - 1;
    uint32_t const step = (tableSize >> 1) + (tableSize >> 3) + 3;               /* FSE_TABLESTEP */
    uint16_t cumul[66]; uint8_t sym[512];
    uint32_t high = tableSize - 1, u, pos = 0; unsigned s, total = 0;
    ct->tableLog = tableLog; ct->maxSym = maxSym;
    cumul[0] = 0;
    for (u = 1; u <= maxSym + 1; u++) {
        if (norm[u-1] == -1) { cumul[u] = cumul[u-1] + 1; sym[high--] = (uint8_t)(u - 1); }
        else cumul[u] = cumul[u-1] + (uint16_t)norm[u-1];
    }
    for (s = 0; s <= maxSym; s++) {
        int i;
        for (i = 0; i < norm[s]; i++) {
            sym[pos] = (uint8_t)s;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    }
    for (u = 0; u < tableSize; u++) { uint8_t const c = sym[u]; ct->state[cumul[c]++] = (uint16_t)(tableSize + u); }
    for (s = 0; s <= maxSym; s++) {
        switch (norm[s]) {
        case 0: ct->dBits[s] = ((tableLog + 1) << 16) - (1u << tableLog); ct->dFind[s] = 0; break;
        case -1: case 1:
            ct->dBits[s] = (tableLog << 16) - (1u << tableLog);
            ct->dFind[s] = (int)(total - 1); total++; break;
        default: {
            uint32_t const maxBitsOut = tableLog - hb32((uint32_t)norm[s] - 1);
            uint32_t const minStatePlus = (uint32_t)norm[s] << maxBitsOut;
            ct->dBits[s] = (maxBitsOut << 16) - minStatePlus;
            ct->dFind[s] = (int)(total - (unsigned)norm[s]);
            total += (unsigned)norm[s]; }
        }
    }
}
static void fse_build_rle(zo_fse* ct, unsigned symbol)                           /* fse_compress.c:528 */
{
    ct->tableLog = 0; ct->state[0] = 0; ct->state[1] = 0; ct->maxSym = symbol;
    ct->dBits[symbol] = 0; ct->dFind[symbol] = 0;
}

/* lib/common/fse.h:452-476 */
static uint32_t fse_init2(const zo_fse* ct, unsigned symbol)
{
    uint32_t const nbBitsOut = (ct->dBits[symbol] + (1 << 15)) >> 16;
    uint32_t const v = (nbBitsOut << 16) - ct->dBits[symbol];
    return ct->state[(v >> nbBitsOut) + ct->dFind[symbol]];
}
static uint32_t fse_encode(zo_bits* b, const zo_fse* ct, uint32_t state, unsigned symbol)
{
    uint32_t const nbBitsOut = (state + ct->dBits[symbol]) >> 16;
    bw_add(b, state, nbBitsOut);
    return ct->state[(state >> nbBitsOut) + ct->dFind[symbol]];
}

/* ------------------------------------------------------------------ Huffman */
typedef struct { uint32_t count; uint16_t parent; uint8_t byte; uint8_t nbBits; } zo_node;

/* huf_compress.c:524-533: RANK_POSITION_DISTINCT_COUNT_CUTOFF = 158 + highbit32(158) = 165 (the reference's comment says 166) */
static unsigned huf_bucket(uint32_t c) { return c < 165 ? c : hb32(c) + 158; }

static void huf_isort(zo_node* a, int low, int high)                             /* :555 */
{
    int i, size = high - low + 1; a += low;
    for (i = 1; i < size; i++) {
        zo_node const key = a[i]; int j = i - 1;
        while (j >= 0 && a[j].count < key.count) { a[j+1] = a[j]; j--; }
        a[j+1] = key;
    }
}
static int huf_partition(zo_node* a, int low, int high)                          /* :571 */
{
    uint32_t const pivot = a[high].count; int i = low - 1, j; zo_node t;
    for (j = low; j < high; j++) if (a[j].count > pivot) { i++; t = a[i]; a[i] = a[j]; a[j] = t; }
    t = a[i+1]; a[i+1] = a[high]; a[high] = t;
    return i + 1;
}
static void huf_qsort(zo_node* a, int low, int high)                             /* :591 */
{
    if (high - low < 8) { huf_isort(a, low, high); return; }
    while (low < high) {
        int const idx = huf_partition(a, low, high);
        if (idx - low < high - idx) { huf_qsort(a, low, idx - 1); low = idx + 1; }
        else { huf_qsort(a, idx + 1, high); high = idx - 1; }
    }
}

/* huf_compress.c:620-665 */
static void huf_sort(zo_node* node, const unsigned* count, unsigned maxSym)
{
    struct { uint16_t base, curr; } rp[192];
    unsigned n;
    memset(rp, 0, sizeof(rp));
    for (n = 0; n <= maxSym; n++) rp[huf_bucket(count[n])].base++;
    for (n = 191; n > 0; n--) { rp[n-1].base += rp[n].base; rp[n-1].curr = rp[n-1].base; }
    for (n = 0; n <= maxSym; n++) {
        unsigned const r = huf_bucket(count[n]) + 1;
        unsigned const pos = rp[r].curr++;
        node[pos].count = count[n]; node[pos].byte = (uint8_t)n;
    }
    for (n = 165; n < 191; n++) {       /* from the cutoff: rp[165] holds the symbols whose count is exactly 164 */
        int const sz = rp[n].curr - rp[n].base;
        if (sz > 1) huf_qsort(node + rp[n].base, 0, sz - 1);
    }
}

/* huf_compress.c:376-498 */
static unsigned huf_set_max_height(zo_node* node, unsigned lastNonNull, unsigned target)
{
    unsigned const largestBits = node[lastNonNull].nbBits;
    if (largestBits <= target) return largestBits;
    {   int totalCost = 0, n = (int)lastNonNull;
        unsigned const baseCost = 1u << (largestBits - target);
        uint32_t rankLast[14]; unsigned const noSymbol = 0xF0F0F0F0;
        while (node[n].nbBits > target) {
            totalCost += (int)(baseCost - (1u << (largestBits - node[n].nbBits)));
            node[n].nbBits = (uint8_t)target; n--;
        }
        while (node[n].nbBits == target) --n;
        totalCost >>= (largestBits - target);
        {   unsigned i; for (i = 0; i < 14; i++) rankLast[i] = noSymbol; }
        {   unsigned cur = target; int pos;
            for (pos = n; pos >= 0; pos--) {
                if (node[pos].nbBits >= cur) continue;
                cur = node[pos].nbBits;
                rankLast[target - cur] = (uint32_t)pos;
        }   }
        while (totalCost > 0) {
            unsigned nBitsToDecrease = hb32((uint32_t)totalCost) + 1;
            for ( ; nBitsToDecrease > 1; nBitsToDecrease--) {
                uint32_t const highPos = rankLast[nBitsToDecrease], lowPos = rankLast[nBitsToDecrease - 1];
                if (highPos == noSymbol) continue;
                if (lowPos == noSymbol) break;
                if (node[highPos].count <= 2 * node[lowPos].count) break;
            }
            while (nBitsToDecrease <= 12 && rankLast[nBitsToDecrease] == noSymbol) nBitsToDecrease++;
            totalCost -= 1 << (nBitsToDecrease - 1);
            node[rankLast[nBitsToDecrease]].nbBits++;
            if (rankLast[nBitsToDecrease - 1] == noSymbol) rankLast[nBitsToDecrease - 1] = rankLast[nBitsToDecrease];
            if (rankLast[nBitsToDecrease] == 0) rankLast[nBitsToDecrease] = noSymbol;
            else {
                rankLast[nBitsToDecrease]--;
                if (node[rankLast[nBitsToDecrease]].nbBits != target - nBitsToDecrease) rankLast[nBitsToDecrease] = noSymbol;
            }
        }
        while (totalCost < 0) {
            if (rankLast[1] == noSymbol) {
                while (node[n].nbBits == target) n--;
                node[n+1].nbBits--; rankLast[1] = (uint32_t)(n + 1); totalCost++;
                continue;
            }
            node[rankLast[1] + 1].nbBits--; rankLast[1]++; totalCost++;
        }
    }
    return target;
}

/* huf_compress.c:756-791 (+ :681-718 tree, :730-753 canonical values). value[] = code, nbBits[] = length */
static unsigned huf_build_full(const unsigned* count, unsigned maxSym, unsigned maxNbBits, uint8_t nbBits[256], uint16_t value[256])
{
    zo_node tbl[514]; zo_node* const node0 = tbl; zo_node* const node = tbl + 1;
    int nonNull, lowS, lowN, nodeNb = 256, nodeRoot, n;
    memset(tbl, 0, sizeof(tbl));
    huf_sort(node, count, maxSym);
    nonNull = (int)maxSym;
    while (node[nonNull].count == 0) nonNull--;
    lowS = nonNull; nodeRoot = nodeNb + lowS - 1; lowN = nodeNb;
    node[nodeNb].count = node[lowS].count + node[lowS-1].count;
    node[lowS].parent = node[lowS-1].parent = (uint16_t)nodeNb;
    nodeNb++; lowS -= 2;
    for (n = nodeNb; n <= nodeRoot; n++) node[n].count = 1u << 30;
    node0[0].count = 1u << 31;
    while (nodeNb <= nodeRoot) {
        int const n1 = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
        int const n2 = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
        node[nodeNb].count = node[n1].count + node[n2].count;
        node[n1].parent = node[n2].parent = (uint16_t)nodeNb;
        nodeNb++;
    }
    node[nodeRoot].nbBits = 0;
    for (n = nodeRoot - 1; n >= 256; n--) node[n].nbBits = node[node[n].parent].nbBits + 1;
    for (n = 0; n <= nonNull; n++) node[n].nbBits = node[node[n].parent].nbBits + 1;
    maxNbBits = huf_set_max_height(node, (unsigned)nonNull, maxNbBits);
    {   uint16_t nbPerRank[13] = {0}, valPerRank[13] = {0}, min = 0;
        for (n = 0; n <= nonNull; n++) nbPerRank[node[n].nbBits]++;
        for (n = (int)maxNbBits; n > 0; n--) { valPerRank[n] = min; min += nbPerRank[n]; min >>= 1; }
        memset(nbBits, 0, 256);
        for (n = 0; n <= (int)maxSym; n++) nbBits[node[n].byte] = node[n].nbBits;
        for (n = 0; n <= (int)maxSym; n++) value[n] = nbBits[n] ? valPerRank[nbBits[n]]++ : 0;
    }
    return maxNbBits;
}
unsigned zo_huf_build(const unsigned* count, unsigned maxSym, unsigned maxNbBits, uint8_t nbBits[256])
{
    uint16_t value[256];
    return huf_build_full(count, maxSym, maxNbBits, nbBits, value);
}

/* huf_compress.c:147-186 HUF_compressWeights; fse_compress.c:551-608 for the 2-state coder. returns 0 = not compressible */
static size_t huf_compress_weights(uint8_t* dst, const uint8_t* w, size_t n)
{
    unsigned count[13], maxSym = 12, tableLog;
    short norm[13]; zo_fse ct; uint8_t* op = dst;
    if (n <= 1) return 0;
    if (n == 2) return w[0] == w[1];            /* maxCount == wtSize -> 1, maxCount == 1 -> 0 */
    {   unsigned const maxCount = (unsigned)hist_small(count, &maxSym, w, n);
        if (maxCount == n) return 1;
        if (maxCount == 1) return 0;
    }
    tableLog = fse_optimal_log(6, n, maxSym, 2);
    if (zo_fse_normalize(norm, tableLog, count, n, maxSym, 0) < 0) return 0;
    {   size_t const h = fse_write_ncount(op, norm, maxSym, tableLog);
        if (!h) return 0;
        op += h;
    }
    fse_build(&ct, norm, maxSym, tableLog);
    {   zo_bits b; size_t i = n; uint32_t s1, s2;      /* i = symbols not yet encoded; encode order is last -> first */
        bw_init(&b, op);
        if (n & 1) { s1 = fse_init2(&ct, w[i-1]); s2 = fse_init2(&ct, w[i-2]); s1 = fse_encode(&b, &ct, s1, w[i-3]); i -= 3; }
        else       { s2 = fse_init2(&ct, w[i-1]); s1 = fse_init2(&ct, w[i-2]); i -= 2; }
        if ((n - 2) & 2) { s2 = fse_encode(&b, &ct, s2, w[i-1]); s1 = fse_encode(&b, &ct, s1, w[i-2]); i -= 2; }
        while (i >= 4) {
            s2 = fse_encode(&b, &ct, s2, w[i-1]); s1 = fse_encode(&b, &ct, s1, w[i-2]);
            s2 = fse_encode(&b, &ct, s2, w[i-3]); s1 = fse_encode(&b, &ct, s1, w[i-4]);
            i -= 4;
        }
        bw_add(&b, s2, ct.tableLog); bw_add(&b, s1, ct.tableLog);
        op = bw_close(&b);
    }
    return (size_t)(op - dst);
}

/* huf_compress.c:248-289 HUF_writeCTable_wksp; returns size, 0 on failure */
static size_t huf_write_table(uint8_t* dst, const uint8_t nbBits[256], unsigned maxSym, unsigned huffLog)
{
    uint8_t w[256]; unsigned n;
    for (n = 0; n < maxSym; n++) w[n] = nbBits[n] ? (uint8_t)(huffLog + 1 - nbBits[n]) : 0;
    {   size_t const h = huf_compress_weights(dst + 1, w, maxSym);
        if (h > 1 && h < maxSym / 2) { dst[0] = (uint8_t)h; return h + 1; }
    }
    if (maxSym > 128) return 0;
    dst[0] = (uint8_t)(128 + (maxSym - 1));
    w[maxSym] = 0;
    for (n = 0; n < maxSym; n += 2) dst[n/2 + 1] = (uint8_t)((w[n] << 4) + w[n+1]);
    return ((maxSym + 1) / 2) + 1;
}

/* huf_compress.c:1056-1118: one stream, symbols last->first, closed by a 1 bit (bytes independent of flush policy) */
static size_t huf_encode_1x(uint8_t* dst, const uint8_t* src, size_t n, const uint8_t* nbBits, const uint16_t* value)
{
    zo_bits b; size_t i;
    bw_init(&b, dst);
    for (i = n; i-- > 0; ) bw_add(&b, value[src[i]], nbBits[src[i]]);
    return (size_t)(bw_close(&b) - dst);
}
/* huf_compress.c:1168-1215 */
static size_t huf_encode_4x(uint8_t* dst, const uint8_t* src, size_t n, const uint8_t* nbBits, const uint16_t* value)
{
    size_t const seg = (n + 3) / 4; uint8_t* op = dst + 6; int k;
    if (n < 12) return 0;
    for (k = 0; k < 4; k++) {
        size_t const len = (k < 3) ? seg : n - 3 * seg;
        size_t const c = huf_encode_1x(op, src + (size_t)k * seg, len, nbBits, value);
        if (c == 0 || c > 65535) return 0;
        if (k < 3) wr16(dst + 2 * k, (unsigned)c);
        op += c;
    }
    return (size_t)(op - dst);
}

/* huf_compress.c:1333-1434 HUF_compress_internal with no previous table.
 * returns compressed size (table + streams), 0 = not compressible, 1 = single symbol (dst[0] = symbol) */
static size_t huf_compress(uint8_t* dst, const uint8_t* src, size_t n, int fourStreams, int suspect)
{
    unsigned count[256], maxSym = 255, huffLog; uint8_t nbBits[256]; uint16_t value[256];
    uint8_t* op = dst;
    if (!n) return 0;
    if (suspect && n >= 4096 * 10) {                                             /* :1367-1379 */
        unsigned c2[256], m2; size_t tot;
        tot = zo_hist(c2, &m2, src, 4096);
        tot += zo_hist(c2, &m2, src + n - 4096, 4096);
        if (tot <= ((2 * 4096) >> 7) + 4) return 0;
    }
    {   size_t const largest = zo_hist(count, &maxSym, src, n);                  /* :1382-1385 */
        if (largest == n) { dst[0] = src[0]; return 1; }
        if (largest <= (n >> 7) + 4) return 0;
    }
    huffLog = fse_optimal_log(11, n, maxSym, 1);                                 /* :1284-1287 */
    huffLog = huf_build_full(count, maxSym, huffLog, nbBits, value);             /* :1403-1409 */
    {   size_t const h = huf_write_table(op, nbBits, maxSym, huffLog);           /* :1412-1430 */
        if (!h) return ZO_ERROR;
        if (h + 12 >= n) return 0;
        op += h;
    }
    {   size_t const c = fourStreams ? huf_encode_4x(op, src, n, nbBits, value)
                                     : huf_encode_1x(op, src, n, nbBits, value); /* :1224-1239 */
        if (c == 0) return 0;
        op += c;
        if ((size_t)(op - dst) >= n - 1) return 0;
    }
    return (size_t)(op - dst);
}

/* ------------------------------------------------------------------ previous-block entropy state (dictionaries)
 * What a ZDICT-format dictionary puts into the CDict's block state (zstd_compress.c:4986-5076 ZSTD_loadCEntropy): a Huffman
 * table for literals, FSE tables for offset / match-length / literal-length codes, each with a repeat mode (1 = "check":
 * usable only if it covers the block's symbols, 2 = "valid": covers every symbol). */
typedef struct {
    int hufRepeat; unsigned hufMaxSym; uint8_t hufNbBits[256]; uint16_t hufValue[256];
    int llRepeat, ofRepeat, mlRepeat;
    zo_fse ll, of, ml;
} zo_prev;

/* LSB-first forward bit reader for headers */
typedef struct { const uint8_t* p; size_t size; size_t bit; } zo_fbits;
static uint32_t fb_peek(const zo_fbits* b, unsigned n)
{
    uint64_t v = 0; size_t const byte = b->bit >> 3; unsigned i;
    for (i = 0; i < 8; i++) if (byte + i < b->size) v |= (uint64_t)b->p[byte + i] << (8 * i);
    return (uint32_t)((v >> (b->bit & 7)) & ((1ULL << n) - 1));
}
/* lib/common/entropy_common.c:42-214 FSE_readNCount (format: doc/zstd_compression_format.md "FSE Table Description").
 * returns bytes consumed, 0 on error */
static size_t fse_read_ncount(short* norm, unsigned* maxSym, unsigned* tableLog, const uint8_t* src, size_t size)
{
    zo_fbits b; int remaining, threshold, nbBits; unsigned charnum = 0, maxSV1 = *maxSym + 1; int previous0 = 0;
    b.p = src; b.size = size; b.bit = 0;
    memset(norm, 0, sizeof(short) * maxSV1);
    nbBits = (int)fb_peek(&b, 4) + 5; b.bit += 4;
    if (nbBits > 15) return 0;
    *tableLog = (unsigned)nbBits;
    remaining = (1 << nbBits) + 1; threshold = 1 << nbBits; nbBits++;
    while (remaining > 1 && charnum < maxSV1) {
        if (previous0) {
            for (;;) { uint32_t const r = fb_peek(&b, 2); b.bit += 2; charnum += r; if (r != 3) break; }
            if (charnum >= maxSV1) break;
        }
        {   int const max = (2 * threshold - 1) - remaining;
            uint32_t const bits = fb_peek(&b, (unsigned)nbBits);
            int count;
            if ((int)(bits & (uint32_t)(threshold - 1)) < max) { count = (int)(bits & (uint32_t)(threshold - 1)); b.bit += (size_t)nbBits - 1; }
            else { count = (int)(bits & (uint32_t)(2 * threshold - 1)); if (count >= threshold) count -= max; b.bit += (size_t)nbBits; }
            count--;
            remaining -= count < 0 ? -count : count;
            norm[charnum++] = (short)count;
            previous0 = !count;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
    }
    if (remaining != 1 || charnum > maxSV1) return 0;
    *maxSym = charnum - 1;
    return (b.bit + 7) >> 3;
}

/* nb bits of a backward bitstream whose lowest bit sits at position `at` (positions below 0 read as 0) */
static unsigned bs_bits(const uint8_t* bs, long at, unsigned nb)
{
    unsigned v = 0, k;
    for (k = 0; k < nb; k++) { long const q = at + (long)k; if (q >= 0 && ((bs[q >> 3] >> (q & 7)) & 1)) v |= 1u << k; }
    return v;
}

/* FSE decoding of Huffman weights (lib/common/fse_decompress.c:58-277; format doc "FSE Decoding"): dst receives at most cap
 * symbols; returns their number, 0 on error.  The bitstream is read backwards from its end mark. */
static size_t fse_decompress_weights(uint8_t* dst, size_t cap, const uint8_t* src, size_t size)
{
    short norm[256]; unsigned maxSym = 255, tl; size_t const h = fse_read_ncount(norm, &maxSym, &tl, src, size);
    uint8_t symT[64]; uint8_t nbT[64]; uint16_t newT[64]; unsigned next[256];
    if (!h || tl > 6 || h >= size) return 0;
    {   unsigned const tsz = 1u << tl, mask = tsz - 1, step = (tsz >> 1) + (tsz >> 3) + 3; unsigned high = tsz - 1, pos = 0, sy, u;
        for (sy = 0; sy <= maxSym; sy++) { if (norm[sy] == -1) { symT[high--] = (uint8_t)sy; next[sy] = 1; } else next[sy] = (unsigned)norm[sy]; }
        for (sy = 0; sy <= maxSym; sy++) { int i; for (i = 0; i < norm[sy]; i++) { symT[pos] = (uint8_t)sy; pos = (pos + step) & mask; while (pos > high) pos = (pos + step) & mask; } }
        if (pos != 0) return 0;
        for (u = 0; u < tsz; u++) { unsigned const ns = next[symT[u]]++; nbT[u] = (uint8_t)(tl - hb32(ns)); newT[u] = (uint16_t)((ns << nbT[u]) - tsz); }
    }
    {   const uint8_t* const bs = src + h; size_t const bsz = size - h;
        long cursor; size_t n = 0; unsigned s1, s2; int which = 0;
        if (bs[bsz - 1] == 0) return 0;
        cursor = (long)(8 * (bsz - 1) + hb32(bs[bsz - 1]));                       /* position of the end mark; bits below it are data */
        cursor -= (long)tl; s1 = bs_bits(bs, cursor, tl);
        cursor -= (long)tl; s2 = bs_bits(bs, cursor, tl);
        for (;;) {
            unsigned* const st = which ? &s2 : &s1; unsigned const other = which ? s1 : s2;
            unsigned const nb = nbT[*st];
            if (n + 2 > cap) return 0;
            dst[n++] = symT[*st];
            cursor -= (long)nb;
            *st = newT[*st] + bs_bits(bs, cursor, nb);
            if (cursor < 0) { dst[n++] = symT[other]; break; }                   /* BIT_DStream_overflow: the other state holds the last symbol */
            which ^= 1;
        }
        return n;
    }
}

/* huf_compress.c:291-339 HUF_readCTable + entropy_common.c:248-320 HUF_readStats: table description -> code lengths/values.
 * returns bytes consumed (0 on error) */
static size_t huf_read_table(zo_prev* pv, const uint8_t* src, size_t size)
{
    uint8_t w[256]; unsigned rank[16], nbSym, tableLog, n; size_t iSize, oSize; uint32_t total = 0;
    if (!size) return 0;
    iSize = src[0];
    if (iSize >= 128) {
        oSize = iSize - 127; iSize = (oSize + 1) / 2;
        if (iSize + 1 > size || oSize >= 256) return 0;
        for (n = 0; n < oSize; n += 2) { w[n] = src[1 + n / 2] >> 4; w[n + 1] = src[1 + n / 2] & 15; }
    } else {
        if (iSize + 1 > size) return 0;
        oSize = fse_decompress_weights(w, 255, src + 1, iSize);
        if (!oSize) return 0;
    }
    memset(rank, 0, sizeof(rank));
    for (n = 0; n < oSize; n++) { if (w[n] > 12) return 0; rank[w[n]]++; total += (1u << w[n]) >> 1; }
    if (!total) return 0;
    tableLog = hb32(total) + 1;
    if (tableLog > 12) return 0;
    {   uint32_t const rest = (1u << tableLog) - total; unsigned const last = hb32(rest) + 1;
        if ((1u << hb32(rest)) != rest) return 0;
        w[oSize] = (uint8_t)last; rank[last]++;
    }
    if (rank[1] < 2 || (rank[1] & 1)) return 0;
    nbSym = (unsigned)oSize + 1;
    pv->hufMaxSym = nbSym - 1;
    pv->hufRepeat = (rank[0] == 0 && nbSym == 256) ? 2 : 1;                     /* zstd_compress.c:5002-5006 */
    memset(pv->hufNbBits, 0, 256); memset(pv->hufValue, 0, sizeof(pv->hufValue));
    for (n = 0; n < nbSym; n++) pv->hufNbBits[n] = w[n] ? (uint8_t)(tableLog + 1 - w[n]) : 0;
    {   uint16_t nbPerRank[16] = {0}, valPerRank[16] = {0}; uint16_t min = 0;
        for (n = 0; n < nbSym; n++) nbPerRank[pv->hufNbBits[n]]++;
        for (n = tableLog; n > 0; n--) { valPerRank[n] = min; min = (uint16_t)(min + nbPerRank[n]); min >>= 1; }
        for (n = 0; n < nbSym; n++) pv->hufValue[n] = pv->hufNbBits[n] ? valPerRank[pv->hufNbBits[n]]++ : 0;
    }
    return iSize + 1;
}

/* zstd_compress.c:4966-4981 ZSTD_dictNCountRepeat */
static int dict_ncount_repeat(const short* norm, unsigned dictMax, unsigned maxSym)
{
    unsigned s;
    if (dictMax < maxSym) return 1;
    for (s = 0; s <= maxSym; s++) if (norm[s] == 0) return 1;
    return 2;
}

/* zstd_compress_literals.c:39 / :81 */
static size_t lits_raw(uint8_t* dst, const uint8_t* src, size_t n)
{
    unsigned const fl = 1 + (n > 31) + (n > 4095);
    if (fl == 1) dst[0] = (uint8_t)(0 + (n << 3));
    else if (fl == 2) wr16(dst, (unsigned)(0 + (1 << 2) + (n << 4)));
    else wr32(dst, (uint32_t)(0 + (3 << 2) + (n << 4)));
    memcpy(dst + fl, src, n);
    return n + fl;
}
static size_t lits_rle(uint8_t* dst, const uint8_t* src, size_t n)
{
    unsigned const fl = 1 + (n > 31) + (n > 4095);
    if (fl == 1) dst[0] = (uint8_t)(1 + (n << 3));
    else if (fl == 2) wr16(dst, (unsigned)(1 + (1 << 2) + (n << 4)));
    else wr32(dst, (uint32_t)(1 + (3 << 2) + (n << 4)));
    dst[fl] = src[0];
    return fl + 1;
}

/* zstd_compress_literals.c:129-235 with prevHuf->repeatMode == HUF_repeat_none */
/* huf_compress.c:1224-1239 HUF_compressCTable_internal with a given code */
static size_t huf_encode_with(uint8_t* dst0, uint8_t* op, const uint8_t* src, size_t n, int fourStreams, const uint8_t* nbBits, const uint16_t* value)
{
    size_t const c = fourStreams ? huf_encode_4x(op, src, n, nbBits, value) : huf_encode_1x(op, src, n, nbBits, value);
    if (c == 0) return 0;
    op += c;
    if ((size_t)(op - dst0) >= n - 1) return 0;
    return (size_t)(op - dst0);
}

/* huf_compress.c:1333-1434 HUF_compress_internal WITH a previous table (*repeat: 0 none, 1 check, 2 valid; set to 0 when a new
 * table is emitted) */
static zo_prev* g_huf_next = NULL;   /* when set: receives the Huffman table huf_compress_prev builds and uses (multi-block frames) */
static size_t huf_compress_prev(uint8_t* dst, const uint8_t* src, size_t n, int fourStreams, int suspect,
                                const zo_prev* pv, int* repeat, int preferRepeat)
{
    unsigned count[256], maxSym = 255, huffLog, sy; uint8_t nbBits[256]; uint16_t value[256];
    uint8_t* op = dst;
    if (!n) return 0;
    if (preferRepeat && *repeat == 2) return huf_encode_with(dst, op, src, n, fourStreams, pv->hufNbBits, pv->hufValue);   /* :1359-1363 */
    if (suspect && n >= 4096 * 10) {
        unsigned c2[256], m2; size_t tot;
        tot = zo_hist(c2, &m2, src, 4096);
        tot += zo_hist(c2, &m2, src + n - 4096, 4096);
        if (tot <= ((2 * 4096) >> 7) + 4) return 0;
    }
    {   size_t const largest = zo_hist(count, &maxSym, src, n);
        if (largest == n) { dst[0] = src[0]; return 1; }
        if (largest <= (n >> 7) + 4) return 0;
    }
    if (*repeat == 1) {                                                          /* :1389-1393 HUF_validateCTable (:860-873) */
        int bad = pv->hufMaxSym < maxSym;
        for (sy = 0; sy <= maxSym && !bad; sy++) bad |= (count[sy] != 0) & (pv->hufNbBits[sy] == 0);
        if (bad) *repeat = 0;
    }
    if (preferRepeat && *repeat != 0) return huf_encode_with(dst, op, src, n, fourStreams, pv->hufNbBits, pv->hufValue);   /* :1395-1399 */
    huffLog = fse_optimal_log(11, n, maxSym, 1);
    huffLog = huf_build_full(count, maxSym, huffLog, nbBits, value);
    {   size_t const h = huf_write_table(op, nbBits, maxSym, huffLog);
        if (!h) return ZO_ERROR;
        if (*repeat != 0) {                                                      /* :1415-1421 */
            size_t oldSize = 0, newSize = 0;
            for (sy = 0; sy <= maxSym; sy++) { oldSize += (size_t)pv->hufNbBits[sy] * count[sy]; newSize += (size_t)nbBits[sy] * count[sy]; }
            oldSize >>= 3; newSize >>= 3;
            if (oldSize <= h + newSize || h + 12 >= n) return huf_encode_with(dst, op, src, n, fourStreams, pv->hufNbBits, pv->hufValue);
        }
        if (h + 12 >= n) return 0;
        op += h;
        *repeat = 0;
    }
    if (g_huf_next) { memcpy(g_huf_next->hufNbBits, nbBits, 256); memcpy(g_huf_next->hufValue, value, sizeof(value)); g_huf_next->hufMaxSym = maxSym; g_huf_next->hufRepeat = 1; }
    return huf_encode_with(dst, op, src, n, fourStreams, nbBits, value);
}

size_t zo_compress_literals_prev(uint8_t* dst, size_t cap, const uint8_t* lits, size_t n, const zo_cparams* cp, int suspect, const zo_prev* pv)
{
    size_t const lh = 3 + (n >= 1024) + (n >= 16384);
    int single = n < 256;
    int repeat = pv ? pv->hufRepeat : 0;
    unsigned hType = 2;
    size_t c;
    (void)cap;
    if (cp->strategy == 1 && cp->targetLength > 0) return lits_raw(dst, lits, n);   /* internal.h:621-634 */
    {   int const shift = (9 - (int)cp->strategy) < 3 ? 9 - (int)cp->strategy : 3;  /* :115-127 */
        size_t const mintc = (repeat == 2) ? 6 : ((size_t)8 << shift);
        if (n < mintc) return lits_raw(dst, lits, n);
    }
    if (repeat == 2 && lh == 3) single = 1;                                         /* :170 */
    if (pv) {
        int const preferRepeat = cp->strategy < 4 && n <= 1024;                    /* :165 */
        c = huf_compress_prev(dst + lh, lits, n, !single, suspect, pv, &repeat, preferRepeat);
        if (repeat != 0) hType = 3;                                                 /* :180-184 set_repeat */
    } else c = huf_compress(dst + lh, lits, n, !single, suspect);
    {   size_t const minGain = (n >> 6) + 2;                                        /* internal.h:613 */
        if (c == 0 || c == ZO_ERROR || c >= n - minGain) return lits_raw(dst, lits, n);
    }
    if (c == 1) {                                                                   /* :192-201 */
        size_t i; int same = 1;
        for (i = 1; i < n; i++) if (lits[i] != lits[0]) { same = 0; break; }
        if (n >= 8 || same) return lits_rle(dst, lits, n);
    }
    if (lh == 3) wr24(dst, (uint32_t)(hType + ((uint32_t)(!single) << 2) + ((uint32_t)n << 4) + ((uint32_t)c << 14)));
    else if (lh == 4) wr32(dst, (uint32_t)(hType + (2 << 2) + ((uint32_t)n << 4) + ((uint32_t)c << 18)));
    else { wr32(dst, (uint32_t)(hType + (3 << 2) + ((uint32_t)n << 4) + ((uint32_t)c << 22))); dst[4] = (uint8_t)(c >> 10); }
    return lh + c;
}

size_t zo_compress_literals(uint8_t* dst, size_t cap, const uint8_t* lits, size_t n, const zo_cparams* cp, int suspect)
{
    return zo_compress_literals_prev(dst, cap, lits, n, cp, suspect, NULL);
}

/* ------------------------------------------------------------------ sequences section */
static const uint8_t kLLbits[36] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0, 1,1,1,1,2,2,3,3, 4,6,7,8,9,10,11,12, 13,14,15,16 };
static const uint8_t kMLbits[53] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,
                                     1,1,1,1,2,2,3,3, 4,4,5,7,8,9,10,11, 12,13,14,15,16 };
static const short kLLnorm[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1, 2,2,2,2,2,2,2,2, 2,3,2,1,1,1,1,1, -1,-1,-1,-1 };
static const short kMLnorm[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,
                                   1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
static const short kOFnorm[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };

static unsigned ll_code(uint32_t ll)                                             /* internal.h:520 */
{
    static const uint8_t t[64] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,16,17,17,18,18,19,19,20,20,20,20,21,21,21,21,
        22,22,22,22,22,22,22,22,23,23,23,23,23,23,23,23,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24 };
    return ll > 63 ? hb32(ll) + 19 : t[ll];
}
static unsigned ml_code(uint32_t mlBase)                                         /* internal.h:537 */
{
    static const uint8_t t[128] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,
        32,32,33,33,34,34,35,35,36,36,36,36,37,37,37,37,38,38,38,38,38,38,38,38,39,39,39,39,39,39,39,39,
        40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,
        42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42 };
    return mlBase > 127 ? hb32(mlBase) + 36 : t[mlBase];
}

/* zstd_compress_sequences.c:21-44 */
static const unsigned kInvProbLog256[256] = {
    0,    2048, 1792, 1642, 1536, 1453, 1386, 1329, 1280, 1236, 1197, 1162, 1130, 1100, 1073, 1047,
    1024, 1001, 980,  960,  941,  923,  906,  889,  874,  859,  844,  830,  817,  804,  791,  779,
    768,  756,  745,  734,  724,  714,  704,  694,  685,  676,  667,  658,  650,  642,  633,  626,
    618,  610,  603,  595,  588,  581,  574,  567,  561,  554,  548,  542,  535,  529,  523,  517,
    512,  506,  500,  495,  489,  484,  478,  473,  468,  463,  458,  453,  448,  443,  438,  434,
    429,  424,  420,  415,  411,  407,  402,  398,  394,  390,  386,  382,  377,  373,  370,  366,
    362,  358,  354,  350,  347,  343,  339,  336,  332,  329,  325,  322,  318,  315,  311,  308,
    305,  302,  298,  295,  292,  289,  286,  282,  279,  276,  273,  270,  267,  264,  261,  258,
    256,  253,  250,  247,  244,  241,  239,  236,  233,  230,  228,  225,  222,  220,  217,  215,
    212,  209,  207,  204,  202,  199,  197,  194,  192,  190,  187,  185,  182,  180,  178,  175,
    173,  171,  168,  166,  164,  162,  159,  157,  155,  153,  151,  149,  146,  144,  142,  140,
    138,  136,  134,  132,  130,  128,  126,  123,  121,  119,  117,  115,  114,  112,  110,  108,
    106,  104,  102,  100,  98,   96,   94,   93,   91,   89,   87,   85,   83,   82,   80,   78,
    76,   74,   73,   71,   69,   67,   66,   64,   62,   61,   59,   57,   55,   54,   52,   50,
    49,   47,   46,   44,   42,   41,   39,   37,   36,   34,   33,   31,   30,   28,   26,   25,
    23,   22,   20,   19,   17,   16,   14,   13,   11,   10,   8,    7,    5,    4,    2,    1,
};

static unsigned fse_optimal_log(unsigned maxLog, size_t n, unsigned maxSym, unsigned minus);
static size_t fse_write_ncount(uint8_t* out0, const short* norm, unsigned maxSym, unsigned tableLog);

/* zstd_compress_sequences.c:103-135 ZSTD_fseBitCost: cost in bits of coding `count` with a previous table, (size_t)-1 when it cannot */
static size_t fse_bit_cost(const zo_fse* ct, const unsigned* count, unsigned max)
{
    size_t cost = 0; unsigned s;
    unsigned const tableLog = ct->tableLog, badCost = (tableLog + 1) << 8;
    if (ct->maxSym < max) return (size_t)-1;
    for (s = 0; s <= max; s++) {
        uint32_t const minNbBits = ct->dBits[s] >> 16, threshold = (minNbBits + 1) << 16;      /* lib/common/fse.h:494-509 FSE_bitCost */
        uint32_t const tableSize = 1u << tableLog;
        uint32_t const deltaFromThreshold = threshold - (ct->dBits[s] + tableSize);
        uint32_t const normalized = (deltaFromThreshold << 8) >> tableLog;
        uint32_t const bitCost = (minNbBits + 1) * 256 - normalized;
        if (count[s] == 0) continue;
        if (bitCost >= badCost) return (size_t)-1;
        cost += (size_t)count[s] * bitCost;
    }
    return cost >> 8;
}

/* zstd_compress_sequences.c:157-235 ZSTD_selectEncodingType. *repeatMode (FSE_repeat: 0 none, 1 check, 2 valid) is the state of
 * prevCT coming in and of the table this block leaves behind going out.  returns set_* (0 basic, 1 rle, 2 compressed, 3 repeat) */
static int select_type(const unsigned* count, unsigned max, unsigned maxCount, size_t nbSeq, unsigned fseLog,
                       const short* defNorm, unsigned defaultNormLog, int defaultAllowed, unsigned strategy,
                       const zo_fse* prevCT, int* repeatMode)
{
    if (maxCount == nbSeq) { *repeatMode = 0; return (defaultAllowed && nbSeq <= 2) ? 0 : 1; }
    if (strategy < 4) {                                                          /* :179-204, strategy < ZSTD_lazy */
        if (defaultAllowed) {
            size_t const mult = 10 - strategy;
            size_t const dynMin = (((size_t)1 << defaultNormLog) * mult) >> 3;
            if (*repeatMode == 2 && nbSeq < 1000) return 3;                      /* :187-191 set_repeat with a VALID previous table */
            if (nbSeq < dynMin || maxCount < (nbSeq >> (defaultNormLog - 1))) { *repeatMode = 0; return 0; }
        }
        *repeatMode = 1;
        return 2;
    }
    {   /* :205-231: estimated costs in bits; an impossible choice costs "error" = more than anything */
        size_t basicCost = (size_t)-1, repeatCost = (size_t)-1, ncountCost, compressedCost;
        unsigned s;
        if (defaultAllowed) {                                                    /* :141-155 ZSTD_crossEntropyCost */
            unsigned const shift = 8 - defaultNormLog;
            size_t cost = 0;
            for (s = 0; s <= max; s++) {
                unsigned const normAcc = defNorm[s] != -1 ? (unsigned)defNorm[s] : 1;
                cost += (size_t)count[s] * kInvProbLog256[normAcc << shift];
            }
            basicCost = cost >> 8;
        }
        if (*repeatMode != 0 && prevCT) repeatCost = fse_bit_cost(prevCT, count, max);
        {   short norm[53]; uint8_t wksp[512];                                   /* :70-77 ZSTD_NCountCost */
            unsigned const tableLog = fse_optimal_log(fseLog, nbSeq, max, 2);
            if (zo_fse_normalize(norm, tableLog, count, nbSeq, max, nbSeq >= 2048) < 0) return -1;
            ncountCost = fse_write_ncount(wksp, norm, max, tableLog);
            if (!ncountCost) return -1;
        }
        {   unsigned cost = 0;                                                   /* :83-97 ZSTD_entropyCost */
            for (s = 0; s <= max; s++) {
                unsigned norm = (unsigned)((256 * count[s]) / nbSeq);
                if (count[s] != 0 && norm == 0) norm = 1;
                cost += count[s] * kInvProbLog256[norm];
            }
            compressedCost = (ncountCost << 3) + (cost >> 8);
        }
        if (basicCost <= repeatCost && basicCost <= compressedCost) { *repeatMode = 0; return 0; }   /* :217-222 */
        if (repeatCost <= compressedCost) return 3;                              /* :223-227, the state of the table stays */
        *repeatMode = 1;
        return 2;
    }
}

/* zstd_compress_sequences.c:243-288; returns bytes written to dst (NCount / rle byte), ZO_ERROR on failure */
static size_t build_ctable(uint8_t* dst, zo_fse* ct, unsigned fseLog, int type, unsigned* count, unsigned max,
                           const uint8_t* codes, size_t nbSeq, const short* defNorm, unsigned defLog, unsigned defMax)
{
    if (type == 1) { fse_build_rle(ct, max); dst[0] = codes[0]; return 1; }
    if (type == 0) { fse_build(ct, defNorm, defMax, defLog); return 0; }
    {   short norm[53]; size_t nbSeq1 = nbSeq;
        unsigned const tableLog = fse_optimal_log(fseLog, nbSeq, max, 2);
        if (count[codes[nbSeq-1]] > 1) { count[codes[nbSeq-1]]--; nbSeq1--; }
        if (zo_fse_normalize(norm, tableLog, count, nbSeq1, max, nbSeq1 >= 2048) < 0) return ZO_ERROR;
        {   size_t const h = fse_write_ncount(dst, norm, max, tableLog);
            if (!h) return ZO_ERROR;
            fse_build(ct, norm, max, tableLog);
            return h;
        }
    }
}

/* zstd_compress.c:2934-2997 + :2756-2873 + zstd_compress_sequences.c:291-382 */
static zo_prev* g_fse_next = NULL;   /* when set: receives the three FSE tables the sequences were coded with and their repeat states (multi-block frames) */
size_t zo_compress_sequences_prev(uint8_t* dst, size_t cap, const zo_seq* seqs, size_t nbSeq, const zo_cparams* cp, const zo_prev* pv);
size_t zo_compress_sequences(uint8_t* dst, size_t cap, const zo_seq* seqs, size_t nbSeq, const zo_cparams* cp)
{
    return zo_compress_sequences_prev(dst, cap, seqs, nbSeq, cp, NULL);
}
size_t zo_compress_sequences_prev(uint8_t* dst, size_t cap, const zo_seq* seqs, size_t nbSeq, const zo_cparams* cp, const zo_prev* pv)
{
    uint8_t* op = dst; uint8_t* seqHead;
    uint8_t *llc, *ofc, *mlc; size_t i, lastCount = 0;
    static zo_fse ctLL, ctOF, ctML;
    unsigned count[64], max; int tLL, tOF, tML;
    int rLL = pv ? pv->llRepeat : 0, rOF = pv ? pv->ofRepeat : 0, rML = pv ? pv->mlRepeat : 0;
    (void)cap;
    if (nbSeq < 128) *op++ = (uint8_t)nbSeq;
    else if (nbSeq < 0x7F00) { op[0] = (uint8_t)((nbSeq >> 8) + 0x80); op[1] = (uint8_t)nbSeq; op += 2; }
    else { op[0] = 0xFF; wr16(op + 1, (unsigned)(nbSeq - 0x7F00)); op += 3; }
    if (nbSeq == 0) return (size_t)(op - dst);
    llc = (uint8_t*)malloc(3 * nbSeq); ofc = llc + nbSeq; mlc = ofc + nbSeq;
    for (i = 0; i < nbSeq; i++) {                                                /* zstd_compress.c:2686-2712 */
        llc[i] = (uint8_t)ll_code(seqs[i].litLength);
        ofc[i] = (uint8_t)hb32(seqs[i].offBase);
        mlc[i] = (uint8_t)ml_code(seqs[i].matchLength - 3);
    }
    seqHead = op++;
    {   size_t h, mf;
        max = 35; mf = hist_small(count, &max, llc, nbSeq);
        tLL = select_type(count, max, (unsigned)mf, nbSeq, 9, kLLnorm, 6, 1, cp->strategy, pv ? &pv->ll : NULL, &rLL);
        if (tLL < 0) { free(llc); return ZO_ERROR; }
        if (tLL == 3) { ctLL = pv->ll; h = 0; } else
        h = build_ctable(op, &ctLL, 9, tLL, count, max, llc, nbSeq, kLLnorm, 6, 35);
        if (h == ZO_ERROR) { free(llc); return ZO_ERROR; }
        if (tLL == 2) lastCount = h;
        op += h;
        max = 31; mf = hist_small(count, &max, ofc, nbSeq);
        tOF = select_type(count, max, (unsigned)mf, nbSeq, 8, kOFnorm, 5, max <= 28, cp->strategy, pv ? &pv->of : NULL, &rOF);
        if (tOF < 0) { free(llc); return ZO_ERROR; }
        if (tOF == 3) { ctOF = pv->of; h = 0; } else
        h = build_ctable(op, &ctOF, 8, tOF, count, max, ofc, nbSeq, kOFnorm, 5, 28);
        if (h == ZO_ERROR) { free(llc); return ZO_ERROR; }
        if (tOF == 2) lastCount = h;
        op += h;
        max = 52; mf = hist_small(count, &max, mlc, nbSeq);
        tML = select_type(count, max, (unsigned)mf, nbSeq, 9, kMLnorm, 6, 1, cp->strategy, pv ? &pv->ml : NULL, &rML);
        if (tML < 0) { free(llc); return ZO_ERROR; }
        if (tML == 3) { ctML = pv->ml; h = 0; } else
        h = build_ctable(op, &ctML, 9, tML, count, max, mlc, nbSeq, kMLnorm, 6, 52);
        if (h == ZO_ERROR) { free(llc); return ZO_ERROR; }
        if (tML == 2) lastCount = h;
        op += h;
    }
    *seqHead = (uint8_t)((tLL << 6) + (tOF << 4) + (tML << 2));
    if (g_fse_next) {                                                            /* nextEntropy->fse: the tables this block used, and their states */
        g_fse_next->ll = ctLL; g_fse_next->of = ctOF; g_fse_next->ml = ctML;
        g_fse_next->llRepeat = rLL; g_fse_next->ofRepeat = rOF; g_fse_next->mlRepeat = rML;
    }
    {   zo_bits b; uint32_t sML, sOF, sLL; size_t n = nbSeq - 1; uint8_t* end;     /* zstd_compress_sequences.c:291-382 */
        bw_init(&b, op);
        sML = fse_init2(&ctML, mlc[n]); sOF = fse_init2(&ctOF, ofc[n]); sLL = fse_init2(&ctLL, llc[n]);
        bw_add(&b, seqs[n].litLength, kLLbits[llc[n]]);
        bw_add(&b, seqs[n].matchLength - 3, kMLbits[mlc[n]]);
        bw_add(&b, seqs[n].offBase, ofc[n]);
        while (n-- > 0) {
            sOF = fse_encode(&b, &ctOF, sOF, ofc[n]);
            sML = fse_encode(&b, &ctML, sML, mlc[n]);
            sLL = fse_encode(&b, &ctLL, sLL, llc[n]);
            bw_add(&b, seqs[n].litLength, kLLbits[llc[n]]);
            bw_add(&b, seqs[n].matchLength - 3, kMLbits[mlc[n]]);
            bw_add(&b, seqs[n].offBase, ofc[n]);
        }
        bw_add(&b, sML, ctML.tableLog); bw_add(&b, sOF, ctOF.tableLog); bw_add(&b, sLL, ctLL.tableLog);
        end = bw_close(&b);
        {   size_t const bs = (size_t)(end - op);
            op = end;
            if (lastCount && lastCount + bs < 4) { free(llc); return 0; }         /* zstd_compress.c:2987 */
        }
    }
    free(llc);
    return (size_t)(op - dst);
}

/* ================================================================== dictionary (CDict, attach mode) ==================
 * SURVEY.md §3.4 / §8 rows a7, a9, a11.  A CDict holds the dictionary content, its own tagged hash tables
 * ("short cache": index << 8 | 8-bit tag, zstd_compress_internal.h:1399-1417) built with the CDict's own parameters,
 * and the block state the first block starts from (repcodes; entropy tables for ZDICT-format dictionaries).
 * Index spaces follow the reference: dictionary byte j has index j + 2 (ZSTD_WINDOW_START_INDEX), the attached working
 * context continues at prefixStart = dictLen + 2 (zstd_compress.c:2352-2362), so dictIndexDelta is 0. */
struct zo_cdict_s {
    uint8_t* content; size_t len;          /* dictionary content (what matches may reference) */
    zo_cparams cp;                         /* CDict parameters (ZSTD_cpm_createCDict) */
    uint32_t* tabL; uint32_t* tabS;        /* fast: tabL only (hashLog); dfast: long (hashLog) + short (chainLog) */
    uint32_t dictID; uint32_t rep[3];
    int level;
    size_t fullSize;                       /* size of the dictionary buffer as given (cdict->dictContentSize) */
    int hasEntropy; zo_prev prev;          /* ZDICT-format dictionaries: the entropy tables the first block starts from */
    /* greedy / lazy / lazy2 CDicts: the dictionary's own hash chain (head + chain, reference indices = byte + 2, 0 = empty) or rows (+ tags,
     * unsalted: a CDict's hashSalt is 0, zstd_compress.c:2036-2040), filled by ZSTD_loadDictionaryContent (:4920-4964) */
    uint32_t* lzHead; uint32_t* lzChain; uint32_t* lzRow; uint8_t* lzTag; int lzUseRow; unsigned lzRowLog;
};

static void zo_put_tagged(uint32_t* t, uint32_t hashAndTag, uint32_t index)     /* internal.h:1404 ZSTD_writeTaggedIndex */
{
    t[hashAndTag >> 8] = (index << 8) | (hashAndTag & 0xFF);
}

/* zstd_fast.c:16-49 ZSTD_fillHashTableForCDict / zstd_double_fast.c:18-54 ZSTD_fillDoubleHashTableForCDict (dtlm_full) */
static void zo_cdict_fill(zo_cdict* cd)
{
    const uint8_t* const base = cd->content - 2;              /* index -> byte */
    size_t const endIdx = cd->len + 2;
    unsigned const mls = cd->cp.minMatch;
    size_t ip, first = 2;
    {   /* zstd_compress.c:4888-4896: a dictionary larger than the tables can reasonably index only has its SUFFIX indexed
         * (all of it stays referenceable) */
        unsigned const m = cd->cp.hashLog > cd->cp.chainLog ? cd->cp.hashLog : cd->cp.chainLog;
        size_t const maxDictSize = (size_t)8 << (m < 28 ? m : 28);
        if (cd->len > maxDictSize) first = 2 + (cd->len - maxDictSize);
    }
    if (endIdx - first <= 8) return;                          /* :4903 srcSize <= HASH_READ_SIZE */
    if (cd->cp.strategy == 1) {
        unsigned const hb = cd->cp.hashLog + 8;
        for (ip = first; ip + 3 < (endIdx - 8) + 2; ip += 3) {    /* :37 ip + step < iend + 2 */
            unsigned p;
            zo_put_tagged(cd->tabL, zo_hash(base + ip, hb, mls), (uint32_t)ip);
            for (p = 1; p < 3; p++) {
                uint32_t const ht = zo_hash(base + ip + p, hb, mls);
                if (cd->tabL[ht >> 8] == 0) zo_put_tagged(cd->tabL, ht, (uint32_t)(ip + p));
            }
        }
    } else {
        unsigned const hbL = cd->cp.hashLog + 8, hbS = cd->cp.chainLog + 8;
        for (ip = first; ip + 3 - 1 <= endIdx - 8; ip += 3) { /* :36 */
            unsigned i;
            for (i = 0; i < 3; i++) {
                uint32_t const sm = zo_hash(base + ip + i, hbS, mls), lg = zo_hash(base + ip + i, hbL, 8);
                if (i == 0) zo_put_tagged(cd->tabS, sm, (uint32_t)(ip + i));
                if (i == 0 || cd->tabL[lg >> 8] == 0) zo_put_tagged(cd->tabL, lg, (uint32_t)(ip + i));
            }
        }
    }
}

/* the lazy strategies' CDict tables: every position of the (indexed suffix of the) dictionary up to 8 before its end goes into the hash
 * chain (ZSTD_insertAndFindFirstIndex, zstd_lazy.c:632-662, hashed with the CDict's minMatch as it is) or into the rows (ZSTD_row_update
 * without the skip rule and without salt, :950-958; minMatch capped at 6).  Row matcher when the CDict's windowLog > 14 (:237-253). */
static void zo_cdict_fill_lazy(zo_cdict* cd)
{
    const uint8_t* const base = cd->content - 2;
    size_t const endIdx = cd->len + 2;
    size_t idx, first = 2;
    cd->lzUseRow = g_zo_row_matcher && cd->cp.windowLog > 14;
    cd->lzRowLog = cd->cp.searchLog < 4 ? 4 : (cd->cp.searchLog > 6 ? 6 : cd->cp.searchLog);
    if (cd->lzUseRow) { cd->lzRow = (uint32_t*)calloc((size_t)1 << cd->cp.hashLog, sizeof(uint32_t)); cd->lzTag = (uint8_t*)calloc((size_t)1 << cd->cp.hashLog, 1); }
    else { cd->lzHead = (uint32_t*)calloc((size_t)1 << cd->cp.hashLog, sizeof(uint32_t)); cd->lzChain = (uint32_t*)calloc((size_t)1 << cd->cp.chainLog, sizeof(uint32_t)); }
    {   unsigned const m = cd->cp.hashLog > cd->cp.chainLog ? cd->cp.hashLog : cd->cp.chainLog;
        size_t const maxDictSize = (size_t)8 << (m < 28 ? m : 28);
        if (cd->len > maxDictSize) first = 2 + (cd->len - maxDictSize);
    }
    if (endIdx - first <= 8) return;
    for (idx = first; idx < endIdx - 8; idx++) {
        if (cd->lzUseRow) {
            unsigned const rowMask = (1u << cd->lzRowLog) - 1, mls = cd->cp.minMatch > 6 ? 6 : cd->cp.minMatch;
            uint32_t const h = zo_hash_salted(base + idx, cd->cp.hashLog - cd->lzRowLog + 8, mls, 0);
            size_t const rel = (size_t)(h >> 8) << cd->lzRowLog;
            unsigned const pos = zo_row_next_index(cd->lzTag + rel, rowMask);
            cd->lzTag[rel + pos] = (uint8_t)h; cd->lzRow[rel + pos] = (uint32_t)idx;
        } else {
            uint32_t const h = zo_hash(base + idx, cd->cp.hashLog, cd->cp.minMatch);
            cd->lzChain[idx & ((1u << cd->cp.chainLog) - 1)] = cd->lzHead[h];
            cd->lzHead[h] = (uint32_t)idx;
        }
    }
}

void zo_cdict_free(zo_cdict* cd)
{
    if (!cd) return;
    free(cd->content ? cd->content - 16 : NULL); free(cd->tabL); free(cd->tabS); free(cd->lzHead); free(cd->lzChain); free(cd->lzRow); free(cd->lzTag); free(cd);
}

/* ZSTD_createCDict (zstd_compress.c:5648): parameters for (level, unknown source, dictSize) in createCDict mode,
 * raw-content dictionaries (no ZDICT magic: zstd_compress.c:5138-5148) — repcodes {1,4,8}, no entropy tables, dictID 0 */
zo_cdict* zo_cdict_create(const void* dict, size_t dictSize, int level)
{
    zo_cdict* cd = (zo_cdict*)calloc(1, sizeof(zo_cdict));
    uint8_t* buf;
    if (!cd) return NULL;
    if (zo_get_cparams_mode(level, ZO_SRCSIZE_UNKNOWN, dictSize, 2, &cd->cp) < 0 || cd->cp.strategy > 5) { free(cd); return NULL; }
    cd->level = level == 0 ? 3 : level;
    cd->fullSize = dictSize;
    cd->dictID = 0; cd->rep[0] = 1; cd->rep[1] = 4; cd->rep[2] = 8; cd->hasEntropy = 0;
    if (dictSize < 8) dictSize = 0;                           /* :5130 dictionaries below 8 bytes are ignored */
    if (dictSize >= 8 && rd32((const uint8_t*)dict) == 0xEC30A437U) {
        /* ZSTD_loadZstdDictionary (zstd_compress.c:5087-5118) + ZSTD_loadCEntropy (:4986-5076); the parameters above were
         * computed from the WHOLE dictionary size, like ZSTD_createCDict does */
        const uint8_t* const d0 = (const uint8_t*)dict; const uint8_t* p = d0 + 8; const uint8_t* const dEnd = d0 + dictSize;
        short ofN[32], mlN[53], llN[36]; unsigned ofMax = 31, ofLog, mlMax = 52, mlLog, llMax = 35, llLog; size_t h;
        cd->dictID = rd32(d0 + 4);
        h = huf_read_table(&cd->prev, p, (size_t)(dEnd - p)); if (!h) { free(cd); return NULL; } p += h;
        h = fse_read_ncount(ofN, &ofMax, &ofLog, p, (size_t)(dEnd - p)); if (!h || ofLog > 8) { free(cd); return NULL; } p += h;
        fse_build(&cd->prev.of, ofN, 31, ofLog);                                /* :5016 all offset symbols, MaxOff */
        h = fse_read_ncount(mlN, &mlMax, &mlLog, p, (size_t)(dEnd - p)); if (!h || mlLog > 9) { free(cd); return NULL; } p += h;
        fse_build(&cd->prev.ml, mlN, mlMax, mlLog);
        cd->prev.mlRepeat = dict_ncount_repeat(mlN, mlMax, 52);
        h = fse_read_ncount(llN, &llMax, &llLog, p, (size_t)(dEnd - p)); if (!h || llLog > 9) { free(cd); return NULL; } p += h;
        fse_build(&cd->prev.ll, llN, llMax, llLog);
        cd->prev.llRepeat = dict_ncount_repeat(llN, llMax, 35);
        if (p + 12 > dEnd) { free(cd); return NULL; }
        cd->rep[0] = rd32(p); cd->rep[1] = rd32(p + 4); cd->rep[2] = rd32(p + 8); p += 12;
        {   size_t const contentSize = (size_t)(dEnd - p);
            unsigned offcodeMax = hb32((uint32_t)contentSize + 131072);         /* :5058-5064 */
            cd->prev.ofRepeat = dict_ncount_repeat(ofN, ofMax, offcodeMax < 31 ? offcodeMax : 31);
            if (!cd->rep[0] || !cd->rep[1] || !cd->rep[2] || cd->rep[0] > contentSize || cd->rep[1] > contentSize || cd->rep[2] > contentSize) { free(cd); return NULL; }
            dict = p; dictSize = contentSize;
        }
        cd->hasEntropy = 1;
    }
    buf = (uint8_t*)calloc(dictSize + 64, 1);
    cd->content = buf + 16; cd->len = dictSize;
    memcpy(cd->content, dict, dictSize);
    if (cd->cp.strategy >= 3) { zo_cdict_fill_lazy(cd); return cd; }
    cd->tabL = (uint32_t*)calloc((size_t)1 << cd->cp.hashLog, sizeof(uint32_t));
    cd->tabS = (uint32_t*)calloc((size_t)1 << cd->cp.chainLog, sizeof(uint32_t));
    zo_cdict_fill(cd);
    return cd;
}

/* zstd_compress_internal.h:797 ZSTD_count_2segments: the match may run off the end of the dictionary into the source */
static size_t zo_count_2seg(const uint8_t* ip, const uint8_t* match, const uint8_t* iEnd, const uint8_t* mEnd, const uint8_t* iStart)
{
    const uint8_t* const vEnd = (ip + (mEnd - match) < iEnd) ? ip + (mEnd - match) : iEnd;
    size_t k = 0;
    while (ip + k < vEnd && ip[k] == match[k]) k++;
    if (match + k != mEnd) return k;
    {   size_t j = 0;
        while (ip + k + j < iEnd && ip[k + j] == iStart[j]) j++;
        return k + j;
    }
}
static size_t zo_count_ptr(const uint8_t* ip, const uint8_t* match, const uint8_t* iEnd)
{
    size_t k = 0;
    while (ip + k < iEnd && ip[k] == match[k]) k++;
    return k;
}

/* ---- greedy / lazy / lazy2 with an ATTACHED dictionary (zstd_lazy.c:1516-1779, dictMode = ZSTD_dictMatchState; searches :667-773 and
 * :1141-1340 with their dictMatchState tails).  Reference indices throughout: dictionary byte j = index j + 2, the working context starts
 * at P = dictLen + 2 (dictIndexDelta = 0), source byte i = index P + i.  The working context's own tables are fresh per source. */
typedef struct {
    uint32_t* head; uint32_t* chain; uint32_t* row; uint8_t* tag;
    unsigned hlog, clog, slog, mls, rowLog; int useRow; uint64_t salt;
    uint32_t nextToUpdate; int lazySkipping;
} zo_lzd;

static void zo_lzd_row_insert(zo_lzd* w, const uint8_t* base, uint32_t from, uint32_t to)
{
    unsigned const rowMask = (1u << w->rowLog) - 1;
    for (; from < to; from++) {
        uint32_t const h = zo_hash_salted(base + from, w->hlog - w->rowLog + 8, w->mls, w->salt);
        size_t const rel = (size_t)(h >> 8) << w->rowLog;
        unsigned const pos = zo_row_next_index(w->tag + rel, rowMask);
        w->tag[rel + pos] = (uint8_t)h; w->row[rel + pos] = from;
    }
}

/* one ZSTD_searchMax call at ip (index curr): the working context's own candidates first, then the dictionary's, sharing the attempts */
static size_t zo_lazy_dms_best(zo_lzd* w, const zo_cdict* cd, const uint8_t* src, size_t n, size_t ipPos, uint32_t* offBase)
{
    uint32_t const P = (uint32_t)cd->len + 2;
    const uint8_t* const base = src - P; const uint8_t* const dictBase = cd->content - 2;
    const uint8_t* const ip = src + ipPos, * const iend = src + n, * const dictEnd = cd->content + cd->len;
    uint32_t const curr = P + (uint32_t)ipPos, lowLimit = P;                      /* loadedDictEnd != 0: lowLimit = window.lowLimit = P */
    size_t ml = 4 - 1;
    if (!w->useRow) {
        uint32_t const cmask = (1u << w->clog) - 1, chainSize = 1u << w->clog;
        uint32_t const minChain = curr > chainSize ? curr - chainSize : 0;
        unsigned nbAttempts = 1u << w->slog;
        uint32_t idx = w->nextToUpdate, m;
        while (idx < curr) {                                                     /* :645-653 */
            uint32_t const h = zo_hash(base + idx, w->hlog, w->mls);
            w->chain[idx & cmask] = w->head[h]; w->head[h] = idx; idx++;
            if (w->lazySkipping) break;
        }
        w->nextToUpdate = curr;
        m = w->head[zo_hash(ip, w->hlog, w->mls)];
        for (; m >= lowLimit && nbAttempts > 0; nbAttempts--) {
            const uint8_t* const match = base + m; size_t cur = 0;
            if (rd32(match + ml - 3) == rd32(ip + ml - 3)) cur = zo_count_ptr(ip, match, iend);
            if (cur > ml) { ml = cur; *offBase = (curr - m) + 3; if (ip + cur == iend) break; }
            if (m <= minChain) break;
            m = w->chain[m & cmask];
        }
        {   uint32_t const dChainSize = 1u << cd->cp.chainLog, dmask = dChainSize - 1, dmsSize = P;      /* :742-770 */
            uint32_t const dmsMinChain = dmsSize > dChainSize ? dmsSize - dChainSize : 0;
            m = cd->lzHead[zo_hash(ip, cd->cp.hashLog, w->mls)];
            for (; m >= 2 && nbAttempts > 0; nbAttempts--) {
                const uint8_t* const match = dictBase + m; size_t cur = 0;
                if (rd32(match) == rd32(ip)) cur = zo_count_2seg(ip + 4, match + 4, iend, dictEnd, src) + 4;
                if (cur > ml) { ml = cur; *offBase = (curr - m) + 3; if (ip + cur == iend) break; }
                if (m <= dmsMinChain) break;
                m = cd->lzChain[m & dmask];
            }
        }
        return ml;
    }
    {   unsigned const rowEntries = 1u << w->rowLog, rowMask = rowEntries - 1;
        unsigned const capped = w->slog < w->rowLog ? w->slog : w->rowLog;
        unsigned nbAttempts = 1u << capped, numMatches = 0, k;
        uint32_t buf[64]; uint32_t h;
        uint32_t const dmsHash = zo_hash_salted(ip, cd->cp.hashLog - cd->lzRowLog + 8, w->mls, 0);   /* :1190-1196: the dictionary's rows, unsalted, the WORKING rowLog */
        size_t const dRel = (size_t)(dmsHash >> 8) << w->rowLog;
        if (!w->lazySkipping) {                                                  /* ZSTD_row_update_internal with its skip rule (:916-947) */
            uint32_t idx = w->nextToUpdate;
            if (curr - idx > 384) { zo_lzd_row_insert(w, base, idx, idx + 96); idx = curr - 32; }
            zo_lzd_row_insert(w, base, idx, curr);
        }
        w->nextToUpdate = curr;
        h = zo_hash_salted(ip, w->hlog - w->rowLog + 8, w->mls, w->salt);
        {   size_t const rel = (size_t)(h >> 8) << w->rowLog;
            uint8_t* const tagRow = w->tag + rel; uint32_t* const row = w->row + rel;
            unsigned const head = tagRow[0] & rowMask;
            for (k = 0; k < rowEntries && nbAttempts > 0; k++) {
                unsigned const pos = (head + k) & rowMask;
                if (tagRow[pos] != (uint8_t)h) continue;
                if (pos == 0) continue;
                if (row[pos] < lowLimit) break;
                buf[numMatches++] = row[pos]; nbAttempts--;
            }
            {   unsigned const pos = zo_row_next_index(tagRow, rowMask);
                tagRow[pos] = (uint8_t)h; row[pos] = w->nextToUpdate++; }
        }
        for (k = 0; k < numMatches; k++) {
            const uint8_t* const match = base + buf[k]; size_t cur = 0;
            if (rd32(match + ml - 3) == rd32(ip + ml - 3)) cur = zo_count_ptr(ip, match, iend);
            if (cur > ml) { ml = cur; *offBase = (curr - buf[k]) + 3; if (ip + cur == iend) break; }
        }
        {   const uint8_t* const tagRow = cd->lzTag + dRel; const uint32_t* const row = cd->lzRow + dRel;   /* :1296-1334 */
            unsigned const head = tagRow[0] & rowMask;
            numMatches = 0;
            for (k = 0; k < rowEntries && nbAttempts > 0; k++) {
                unsigned const pos = (head + k) & rowMask;
                if (tagRow[pos] != (uint8_t)dmsHash) continue;
                if (pos == 0) continue;
                if (row[pos] < 2) break;
                buf[numMatches++] = row[pos]; nbAttempts--;
            }
            for (k = 0; k < numMatches; k++) {
                const uint8_t* const match = dictBase + buf[k]; size_t cur = 0;
                if (rd32(match) == rd32(ip)) cur = zo_count_2seg(ip + 4, match + 4, iend, dictEnd, src) + 4;
                if (cur > ml) { ml = cur; *offBase = (curr - buf[k]) + 3; if (ip + cur == iend) break; }
            }
        }
        return ml;
    }
}

static size_t zo_lazy_dms(const zo_cparams* cp, const zo_cdict* cd, const uint8_t* src, size_t n, zo_store* st, uint32_t rep[3], unsigned depth)
{
    zo_lzd w;
    uint32_t const P = (uint32_t)cd->len + 2;
    const uint8_t* const base = src - P; const uint8_t* const dictBase = cd->content - 2;
    const uint8_t* const iend = src + n, * const dictEnd = cd->content + cd->len;
    size_t ilimit, ip = 0, anchor = 0;
    uint32_t off1 = rep[0], off2 = rep[1];
    memset(&w, 0, sizeof(w));
    w.useRow = cd->lzUseRow;                                                     /* the CDict's choice overrides (zstd_compress.c:2322) */
    w.hlog = cp->hashLog; w.clog = cp->chainLog; w.slog = cp->searchLog;
    w.mls = cp->minMatch < 4 ? 4 : (cp->minMatch > 6 ? 6 : cp->minMatch);
    w.rowLog = cp->searchLog < 4 ? 4 : (cp->searchLog > 6 ? 6 : cp->searchLog);
    w.salt = zo_fresh_hash_salt(); w.nextToUpdate = P;
    if (n < (w.useRow ? 16u : 8u)) return n;
    ilimit = n - (w.useRow ? 16 : 8);
    if (w.useRow) { w.row = (uint32_t*)calloc((size_t)1 << w.hlog, sizeof(uint32_t)); w.tag = (uint8_t*)calloc((size_t)1 << w.hlog, 1); }
    else { w.head = (uint32_t*)calloc((size_t)1 << w.hlog, sizeof(uint32_t)); w.chain = (uint32_t*)calloc((size_t)1 << w.clog, sizeof(uint32_t)); }
#define ZO_REPPTR(idx_) ((idx_) < P ? dictBase + (idx_) : base + (idx_))
#define ZO_REPEND(idx_) ((idx_) < P ? dictEnd : iend)
#define ZO_REPOK(idx_)  ((uint32_t)((P - 1) - (idx_)) >= 3)                       /* ZSTD_index_overlap_check */
    while (ip < ilimit) {
        size_t matchLength = 0, start = ip + 1;
        uint32_t offBase = 1;
        int direct = 0;
        {   uint32_t const repIndex = P + (uint32_t)ip + 1 - off1;               /* :1587-1599 */
            const uint8_t* const repMatch = ZO_REPPTR(repIndex);
            if (ZO_REPOK(repIndex) && rd32(repMatch) == rd32(src + ip + 1)) {
                matchLength = zo_count_2seg(src + ip + 1 + 4, repMatch + 4, iend, ZO_REPEND(repIndex), src) + 4;
                if (depth == 0) direct = 1;
            }
        }
        if (!direct) {
            {   uint32_t found = 999999999;
                size_t const ml2 = zo_lazy_dms_best(&w, cd, src, n, ip, &found);
                if (ml2 > matchLength) { matchLength = ml2; start = ip; offBase = found; }
            }
            if (matchLength < 4) {
                size_t const step = ((ip - anchor) >> 8) + 1;
                ip += step;
                w.lazySkipping = step > 8;
                continue;
            }
            if (depth >= 1)
            while (ip < ilimit) {
                ip++;
                {   uint32_t const repIndex = P + (uint32_t)ip - off1;           /* :1640-1654 (no `offBase` guard in this branch) */
                    const uint8_t* const repMatch = ZO_REPPTR(repIndex);
                    if (ZO_REPOK(repIndex) && rd32(repMatch) == rd32(src + ip)) {
                        size_t const mlRep = zo_count_2seg(src + ip + 4, repMatch + 4, iend, ZO_REPEND(repIndex), src) + 4;
                        int const gain2 = (int)(mlRep * 3);
                        int const gain1 = (int)(matchLength * 3 - zo_gain_bits(offBase) + 1);
                        if (mlRep >= 4 && gain2 > gain1) { matchLength = mlRep; offBase = 1; start = ip; }
                    }
                }
                {   uint32_t cand = 999999999;
                    size_t const ml2 = zo_lazy_dms_best(&w, cd, src, n, ip, &cand);
                    int const gain2 = (int)(ml2 * 4 - zo_gain_bits(cand));
                    int const gain1 = (int)(matchLength * 4 - zo_gain_bits(offBase) + 4);
                    if (ml2 >= 4 && gain2 > gain1) { matchLength = ml2; offBase = cand; start = ip; continue; }
                }
                if (depth == 2 && ip < ilimit) {
                    ip++;
                    {   uint32_t const repIndex = P + (uint32_t)ip - off1;
                        const uint8_t* const repMatch = ZO_REPPTR(repIndex);
                        if (ZO_REPOK(repIndex) && rd32(repMatch) == rd32(src + ip)) {
                            size_t const mlRep = zo_count_2seg(src + ip + 4, repMatch + 4, iend, ZO_REPEND(repIndex), src) + 4;
                            int const gain2 = (int)(mlRep * 4);
                            int const gain1 = (int)(matchLength * 4 - zo_gain_bits(offBase) + 1);
                            if (mlRep >= 4 && gain2 > gain1) { matchLength = mlRep; offBase = 1; start = ip; }
                        }
                    }
                    {   uint32_t cand = 999999999;
                        size_t const ml2 = zo_lazy_dms_best(&w, cd, src, n, ip, &cand);
                        int const gain2 = (int)(ml2 * 4 - zo_gain_bits(cand));
                        int const gain1 = (int)(matchLength * 4 - zo_gain_bits(offBase) + 7);
                        if (ml2 >= 4 && gain2 > gain1) { matchLength = ml2; offBase = cand; start = ip; continue; }
                    }
                }
                break;
            }
            if (offBase > 3) {                                                   /* :1716-1722 catch up over both segments */
                uint32_t const matchIndex = P + (uint32_t)start - (offBase - 3);
                const uint8_t* match = ZO_REPPTR(matchIndex);
                const uint8_t* const mStart = matchIndex < P ? cd->content : src;
                while (start > anchor && match > mStart && src[start - 1] == match[-1]) { start--; match--; matchLength++; }
                off2 = off1; off1 = offBase - 3;
            }
        }
        zo_store_seq(st, src, anchor, start - anchor, offBase, (uint32_t)matchLength);
        anchor = ip = start + matchLength;
        w.lazySkipping = 0;
        while (ip <= ilimit) {                                                   /* :1741-1760 */
            uint32_t const repIndex = P + (uint32_t)ip - off2;
            const uint8_t* const repMatch = ZO_REPPTR(repIndex);
            if (ZO_REPOK(repIndex) && rd32(repMatch) == rd32(src + ip)) {
                uint32_t const t = off2;
                matchLength = zo_count_2seg(src + ip + 4, repMatch + 4, iend, ZO_REPEND(repIndex), src) + 4;
                off2 = off1; off1 = t;
                zo_store_seq(st, src, anchor, 0, 1, (uint32_t)matchLength);
                ip += matchLength; anchor = ip;
                continue;
            }
            break;
        }
    }
#undef ZO_REPPTR
#undef ZO_REPEND
#undef ZO_REPOK
    rep[0] = off1; rep[1] = off2;                                                /* no saved offsets in this mode (:1777-1783 with both zero) */
    free(w.head); free(w.chain); free(w.row); free(w.tag);
    return n - anchor;
}

/* ---- greedy / lazy / lazy2 in the dictionary's COPY mode (sources above 32 KB: ZSTD_resetCCtx_byCopyingCDict, zstd_compress.c:2395-2470 —
 * the CDict's hash chain / rows, tags and hash salt (0) are copied, the dictionary becomes the extDict segment; then
 * ZSTD_compressBlock_lazy_extDict_generic, zstd_lazy.c:1937-2137, and the extDict branches of the two searches).  One set of tables:
 * entries below P = dictLen + 2 point into the dictionary, the others into the source. */
static size_t zo_lazy_ext_best(zo_lzd* w, const zo_cdict* cd, const uint8_t* src, size_t n, size_t ipPos, uint32_t* offBase)
{
    uint32_t const P = (uint32_t)cd->len + 2;
    const uint8_t* const base = src - P; const uint8_t* const dictBase = cd->content - 2;
    const uint8_t* const ip = src + ipPos, * const iend = src + n, * const dictEnd = cd->content + cd->len;
    uint32_t const curr = P + (uint32_t)ipPos, lowLimit = 2;                      /* loadedDictEnd != 0: the whole dictionary stays valid */
    size_t ml = 4 - 1;
    if (!w->useRow) {
        uint32_t const cmask = (1u << w->clog) - 1, chainSize = 1u << w->clog;
        uint32_t const minChain = curr > chainSize ? curr - chainSize : 0;
        unsigned nbAttempts = 1u << w->slog;
        uint32_t idx = w->nextToUpdate, m;
        while (idx < curr) {
            uint32_t const h = zo_hash(base + idx, w->hlog, w->mls);
            w->chain[idx & cmask] = w->head[h]; w->head[h] = idx; idx++;
            if (w->lazySkipping) break;
        }
        w->nextToUpdate = curr;
        m = w->head[zo_hash(ip, w->hlog, w->mls)];
        for (; m >= lowLimit && nbAttempts > 0; nbAttempts--) {
            size_t cur = 0;
            if (m >= P) { const uint8_t* const match = base + m; if (rd32(match + ml - 3) == rd32(ip + ml - 3)) cur = zo_count_ptr(ip, match, iend); }
            else { const uint8_t* const match = dictBase + m; if (rd32(match) == rd32(ip)) cur = zo_count_2seg(ip + 4, match + 4, iend, dictEnd, src) + 4; }
            if (cur > ml) { ml = cur; *offBase = (curr - m) + 3; if (ip + cur == iend) break; }
            if (m <= minChain) break;
            m = w->chain[m & cmask];
        }
        return ml;
    }
    {   unsigned const rowEntries = 1u << w->rowLog, rowMask = rowEntries - 1;
        unsigned const capped = w->slog < w->rowLog ? w->slog : w->rowLog;
        unsigned nbAttempts = 1u << capped, numMatches = 0, k;
        uint32_t buf[64]; uint32_t h;
        if (!w->lazySkipping) {
            uint32_t idx = w->nextToUpdate;
            if (curr - idx > 384) { zo_lzd_row_insert(w, base, idx, idx + 96); idx = curr - 32; }
            zo_lzd_row_insert(w, base, idx, curr);
        }
        w->nextToUpdate = curr;
        h = zo_hash_salted(ip, w->hlog - w->rowLog + 8, w->mls, w->salt);
        {   size_t const rel = (size_t)(h >> 8) << w->rowLog;
            uint8_t* const tagRow = w->tag + rel; uint32_t* const row = w->row + rel;
            unsigned const head = tagRow[0] & rowMask;
            for (k = 0; k < rowEntries && nbAttempts > 0; k++) {
                unsigned const pos = (head + k) & rowMask;
                if (tagRow[pos] != (uint8_t)h) continue;
                if (pos == 0) continue;
                if (row[pos] < lowLimit) break;
                buf[numMatches++] = row[pos]; nbAttempts--;
            }
            {   unsigned const pos = zo_row_next_index(tagRow, rowMask);
                tagRow[pos] = (uint8_t)h; row[pos] = w->nextToUpdate++; }
        }
        for (k = 0; k < numMatches; k++) {
            uint32_t const m = buf[k]; size_t cur = 0;
            if (m >= P) { const uint8_t* const match = base + m; if (rd32(match + ml - 3) == rd32(ip + ml - 3)) cur = zo_count_ptr(ip, match, iend); }
            else { const uint8_t* const match = dictBase + m; if (rd32(match) == rd32(ip)) cur = zo_count_2seg(ip + 4, match + 4, iend, dictEnd, src) + 4; }
            if (cur > ml) { ml = cur; *offBase = (curr - m) + 3; if (ip + cur == iend) break; }
        }
        return ml;
    }
}

static size_t zo_lazy_ext(const zo_cparams* cp, const zo_cdict* cd, const uint8_t* src, size_t n, zo_store* st, uint32_t rep[3], unsigned depth)
{
    zo_lzd w;
    uint32_t const P = (uint32_t)cd->len + 2;
    const uint8_t* const base = src - P; const uint8_t* const dictBase = cd->content - 2;
    const uint8_t* const iend = src + n, * const dictEnd = cd->content + cd->len;
    size_t ilimit, ip = 0, anchor = 0;
    uint32_t off1 = rep[0], off2 = rep[1];
    memset(&w, 0, sizeof(w));
    w.useRow = cd->lzUseRow;
    w.hlog = cp->hashLog; w.clog = cp->chainLog; w.slog = cp->searchLog;
    w.mls = cp->minMatch < 4 ? 4 : (cp->minMatch > 6 ? 6 : cp->minMatch);
    w.rowLog = cp->searchLog < 4 ? 4 : (cp->searchLog > 6 ? 6 : cp->searchLog);
    w.salt = 0; w.nextToUpdate = P;                                             /* the CDict's salt and nextToUpdate come along (:2445, :2463) */
    if (n < (w.useRow ? 16u : 8u)) return n;
    ilimit = n - (w.useRow ? 16 : 8);
    if (w.useRow) {
        w.row = (uint32_t*)malloc(sizeof(uint32_t) << w.hlog); w.tag = (uint8_t*)malloc((size_t)1 << w.hlog);
        memcpy(w.row, cd->lzRow, sizeof(uint32_t) << w.hlog); memcpy(w.tag, cd->lzTag, (size_t)1 << w.hlog);
    } else {
        w.head = (uint32_t*)malloc(sizeof(uint32_t) << w.hlog); w.chain = (uint32_t*)malloc(sizeof(uint32_t) << w.clog);
        memcpy(w.head, cd->lzHead, sizeof(uint32_t) << w.hlog); memcpy(w.chain, cd->lzChain, sizeof(uint32_t) << w.clog);
    }
#define ZO_REPPTR(idx_) ((idx_) < P ? dictBase + (idx_) : base + (idx_))
#define ZO_REPEND(idx_) ((idx_) < P ? dictEnd : iend)
#define ZO_REPOK(idx_, off_, cur_) (((uint32_t)((P - 1) - (idx_)) >= 3) & ((off_) <= (cur_) - 2))   /* overlap check & offset <= curr - windowLow, windowLow = 2 */
    ip += 1;                                                                     /* :1967 ip == prefixStart */
    while (ip < ilimit) {
        size_t matchLength = 0, start = ip + 1;
        uint32_t offBase = 1;
        uint32_t curr = P + (uint32_t)ip;
        int direct = 0;
        {   uint32_t const repIndex = curr + 1 - off1;
            if (ZO_REPOK(repIndex, off1, curr + 1) && rd32(src + ip + 1) == rd32(ZO_REPPTR(repIndex))) {
                matchLength = zo_count_2seg(src + ip + 1 + 4, ZO_REPPTR(repIndex) + 4, iend, ZO_REPEND(repIndex), src) + 4;
                if (depth == 0) direct = 1;
            }
        }
        if (!direct) {
            {   uint32_t found = 999999999;
                size_t const ml2 = zo_lazy_ext_best(&w, cd, src, n, ip, &found);
                if (ml2 > matchLength) { matchLength = ml2; start = ip; offBase = found; }
            }
            if (matchLength < 4) {
                size_t const step = (ip - anchor) >> 8;                          /* :2003-2013: the threshold is on the step WITHOUT its + 1 here */
                ip += step + 1;
                w.lazySkipping = step > 8;
                continue;
            }
            if (depth >= 1)
            while (ip < ilimit) {
                ip++; curr++;
                if (offBase) {
                    uint32_t const repIndex = curr - off1;
                    if (ZO_REPOK(repIndex, off1, curr) && rd32(src + ip) == rd32(ZO_REPPTR(repIndex))) {
                        size_t const mlRep = zo_count_2seg(src + ip + 4, ZO_REPPTR(repIndex) + 4, iend, ZO_REPEND(repIndex), src) + 4;
                        int const gain2 = (int)(mlRep * 3);
                        int const gain1 = (int)(matchLength * 3 - zo_gain_bits(offBase) + 1);
                        if (mlRep >= 4 && gain2 > gain1) { matchLength = mlRep; offBase = 1; start = ip; }
                    }
                }
                {   uint32_t cand = 999999999;
                    size_t const ml2 = zo_lazy_ext_best(&w, cd, src, n, ip, &cand);
                    int const gain2 = (int)(ml2 * 4 - zo_gain_bits(cand));
                    int const gain1 = (int)(matchLength * 4 - zo_gain_bits(offBase) + 4);
                    if (ml2 >= 4 && gain2 > gain1) { matchLength = ml2; offBase = cand; start = ip; continue; }
                }
                if (depth == 2 && ip < ilimit) {
                    ip++; curr++;
                    if (offBase) {
                        uint32_t const repIndex = curr - off1;
                        if (ZO_REPOK(repIndex, off1, curr) && rd32(src + ip) == rd32(ZO_REPPTR(repIndex))) {
                            size_t const mlRep = zo_count_2seg(src + ip + 4, ZO_REPPTR(repIndex) + 4, iend, ZO_REPEND(repIndex), src) + 4;
                            int const gain2 = (int)(mlRep * 4);
                            int const gain1 = (int)(matchLength * 4 - zo_gain_bits(offBase) + 1);
                            if (mlRep >= 4 && gain2 > gain1) { matchLength = mlRep; offBase = 1; start = ip; }
                        }
                    }
                    {   uint32_t cand = 999999999;
                        size_t const ml2 = zo_lazy_ext_best(&w, cd, src, n, ip, &cand);
                        int const gain2 = (int)(ml2 * 4 - zo_gain_bits(cand));
                        int const gain1 = (int)(matchLength * 4 - zo_gain_bits(offBase) + 7);
                        if (ml2 >= 4 && gain2 > gain1) { matchLength = ml2; offBase = cand; start = ip; continue; }
                    }
                }
                break;
            }
            if (offBase > 3) {
                uint32_t const matchIndex = P + (uint32_t)start - (offBase - 3);
                const uint8_t* match = ZO_REPPTR(matchIndex);
                const uint8_t* const mStart = matchIndex < P ? cd->content : src;      /* dictStart = dictBase + lowLimit */
                while (start > anchor && match > mStart && src[start - 1] == match[-1]) { start--; match--; matchLength++; }
                off2 = off1; off1 = offBase - 3;
            }
        }
        zo_store_seq(st, src, anchor, start - anchor, offBase, (uint32_t)matchLength);
        anchor = ip = start + matchLength;
        w.lazySkipping = 0;
        while (ip <= ilimit) {
            uint32_t const repCurrent = P + (uint32_t)ip, repIndex = repCurrent - off2;
            if (ZO_REPOK(repIndex, off2, repCurrent) && rd32(src + ip) == rd32(ZO_REPPTR(repIndex))) {
                uint32_t const t = off2;
                matchLength = zo_count_2seg(src + ip + 4, ZO_REPPTR(repIndex) + 4, iend, ZO_REPEND(repIndex), src) + 4;
                off2 = off1; off1 = t;
                zo_store_seq(st, src, anchor, 0, 1, (uint32_t)matchLength);
                ip += matchLength; anchor = ip;
                continue;
            }
            break;
        }
    }
#undef ZO_REPPTR
#undef ZO_REPEND
#undef ZO_REPOK
    rep[0] = off1; rep[1] = off2;
    free(w.head); free(w.chain); free(w.row); free(w.tag);
    return n - anchor;
}

/* zstd_double_fast.c:328-547 ZSTD_compressBlock_doubleFast_dictMatchState_generic */
static size_t zo_dfast_dms(const zo_cparams* cp, const zo_cdict* cd, const uint8_t* src, size_t n, zo_store* st, uint32_t rep[3])
{
    unsigned const hBitsL = cp->hashLog, hBitsS = cp->chainLog, mls = cp->minMatch;
    unsigned const dHBitsL = cd->cp.hashLog + 8, dHBitsS = cd->cp.chainLog + 8;
    uint32_t* const hashLong = (uint32_t*)calloc((size_t)1 << hBitsL, sizeof(uint32_t));
    uint32_t* const hashSmall = (uint32_t*)calloc((size_t)1 << hBitsS, sizeof(uint32_t));
    uint32_t const P = (uint32_t)cd->len + 2;                  /* prefixLowestIndex */
    const uint8_t* const base = src - P;                       /* index -> source byte (index >= P) */
    const uint8_t* const dictBase = cd->content - 2;           /* index -> dictionary byte (index < P) */
    const uint8_t* const dictStart = cd->content, * const dictEnd = cd->content + cd->len;
    const uint8_t* const istart = src, * const iend = src + n, * const ilimit = iend - 8, * const prefixLowest = src;
    const uint8_t* ip = istart, * anchor = istart;
    uint32_t offset_1 = rep[0], offset_2 = rep[1];
#define SEQ(litLen, offBase, ml) zo_store_seq(st, src, (size_t)(anchor - istart), (size_t)(litLen), (offBase), (uint32_t)(ml))
    while (ip < ilimit) {
        size_t mLength; uint32_t offset;
        uint32_t const h2 = zo_hash(ip, hBitsL, 8), h = zo_hash(ip, hBitsS, mls);
        uint32_t const dL = zo_hash(ip, dHBitsL, 8), dS = zo_hash(ip, dHBitsS, mls);
        uint32_t const dEntL = cd->tabL[dL >> 8], dEntS = cd->tabS[dS >> 8];
        int const tagL = (dEntL & 0xFF) == (dL & 0xFF), tagS = (dEntS & 0xFF) == (dS & 0xFF);
        uint32_t const curr = (uint32_t)(ip - base);
        uint32_t const matchIndexL = hashLong[h2];
        uint32_t matchIndexS = hashSmall[h];
        const uint8_t* matchLong = base + matchIndexL;
        const uint8_t* match = base + matchIndexS;
        uint32_t const repIndex = curr + 1 - offset_1;
        const uint8_t* repMatch = repIndex < P ? dictBase + repIndex : base + repIndex;
        hashLong[h2] = hashSmall[h] = curr;
        if ((uint32_t)((P - 1) - repIndex) >= 3 && rd32(repMatch) == rd32(ip + 1)) {        /* :398 ZSTD_index_overlap_check */
            const uint8_t* const repEnd = repIndex < P ? dictEnd : iend;
            mLength = zo_count_2seg(ip + 1 + 4, repMatch + 4, iend, repEnd, prefixLowest) + 4;
            ip++;
            SEQ(ip - anchor, 1, mLength);
            goto _match_stored;
        }
        if (matchIndexL >= P && rd64(matchLong) == rd64(ip)) {                                /* :407 */
            mLength = zo_count_ptr(ip + 8, matchLong + 8, iend) + 8;
            offset = (uint32_t)(ip - matchLong);
            while (ip > anchor && matchLong > prefixLowest && ip[-1] == matchLong[-1]) { ip--; matchLong--; mLength++; }
            goto _match_found;
        } else if (tagL) {                                                                    /* :413 */
            uint32_t const dIdx = dEntL >> 8;
            const uint8_t* dm = dictBase + dIdx;
            if (dm > dictStart && rd64(dm) == rd64(ip)) {
                mLength = zo_count_2seg(ip + 8, dm + 8, iend, dictEnd, prefixLowest) + 8;
                offset = curr - dIdx;
                while (ip > anchor && dm > dictStart && ip[-1] == dm[-1]) { ip--; dm--; mLength++; }
                goto _match_found;
            }
        }
        if (matchIndexS > P) {                                                                /* :427 */
            if (rd32(match) == rd32(ip)) goto _search_next_long;
        } else if (tagS) {
            uint32_t const dIdx = dEntS >> 8;
            match = dictBase + dIdx;
            matchIndexS = dIdx;
            if (match > dictStart && rd32(match) == rd32(ip)) goto _search_next_long;
        }
        ip += ((ip - anchor) >> 8) + 1;                                                      /* :443 */
        continue;
_search_next_long:
        {   uint32_t const hl3 = zo_hash(ip + 1, hBitsL, 8), dL3 = zo_hash(ip + 1, dHBitsL, 8);
            uint32_t const matchIndexL3 = hashLong[hl3], dEntL3 = cd->tabL[dL3 >> 8];
            int const tagL3 = (dEntL3 & 0xFF) == (dL3 & 0xFF);
            const uint8_t* matchL3 = base + matchIndexL3;
            hashLong[hl3] = curr + 1;
            if (matchIndexL3 >= P && rd64(matchL3) == rd64(ip + 1)) {                         /* :459 */
                mLength = zo_count_ptr(ip + 9, matchL3 + 8, iend) + 8;
                ip++;
                offset = (uint32_t)(ip - matchL3);
                while (ip > anchor && matchL3 > prefixLowest && ip[-1] == matchL3[-1]) { ip--; matchL3--; mLength++; }
                goto _match_found;
            } else if (tagL3) {
                uint32_t const dIdx = dEntL3 >> 8;
                const uint8_t* dm = dictBase + dIdx;
                if (dm > dictStart && rd64(dm) == rd64(ip + 1)) {
                    mLength = zo_count_2seg(ip + 1 + 8, dm + 8, iend, dictEnd, prefixLowest) + 8;
                    ip++;
                    offset = curr + 1 - dIdx;
                    while (ip > anchor && dm > dictStart && ip[-1] == dm[-1]) { ip--; dm--; mLength++; }
                    goto _match_found;
                }
            }
        }
        if (matchIndexS < P) {                                                                /* :481 */
            mLength = zo_count_2seg(ip + 4, match + 4, iend, dictEnd, prefixLowest) + 4;
            offset = curr - matchIndexS;
            while (ip > anchor && match > dictStart && ip[-1] == match[-1]) { ip--; match--; mLength++; }
        } else {
            mLength = zo_count_ptr(ip + 4, match + 4, iend) + 4;
            offset = (uint32_t)(ip - match);
            while (ip > anchor && match > prefixLowest && ip[-1] == match[-1]) { ip--; match--; mLength++; }
        }
_match_found:
        offset_2 = offset_1; offset_1 = offset;
        SEQ(ip - anchor, offset + 3, mLength);
_match_stored:
        ip += mLength; anchor = ip;
        if (ip <= ilimit) {                                                                   /* :503 */
            uint32_t const ins = curr + 2;
            hashLong[zo_hash(base + ins, hBitsL, 8)] = ins;
            hashLong[zo_hash(ip - 2, hBitsL, 8)] = (uint32_t)(ip - 2 - base);
            hashSmall[zo_hash(base + ins, hBitsS, mls)] = ins;
            hashSmall[zo_hash(ip - 1, hBitsS, mls)] = (uint32_t)(ip - 1 - base);
            while (ip <= ilimit) {                                                            /* :514 */
                uint32_t const current2 = (uint32_t)(ip - base), repIndex2 = current2 - offset_2;
                const uint8_t* repMatch2 = repIndex2 < P ? dictBase + repIndex2 : base + repIndex2;
                if ((uint32_t)((P - 1) - repIndex2) >= 3 && rd32(repMatch2) == rd32(ip)) {
                    const uint8_t* const repEnd2 = repIndex2 < P ? dictEnd : iend;
                    size_t const rl = zo_count_2seg(ip + 4, repMatch2 + 4, iend, repEnd2, prefixLowest) + 4;
                    uint32_t const t = offset_2; offset_2 = offset_1; offset_1 = t;
                    SEQ(0, 1, rl);
                    hashSmall[zo_hash(ip, hBitsS, mls)] = current2;
                    hashLong[zo_hash(ip, hBitsL, 8)] = current2;
                    ip += rl; anchor = ip;
                    continue;
                }
                break;
            }
        }
    }
    rep[0] = offset_1; rep[1] = offset_2;
    free(hashLong); free(hashSmall);
    return (size_t)(iend - anchor);
}

/* zstd_fast.c:483-678 ZSTD_compressBlock_fast_dictMatchState_generic */
static size_t zo_fast_dms(const zo_cparams* cp, const zo_cdict* cd, const uint8_t* src, size_t n, zo_store* st, uint32_t rep[3])
{
    unsigned const hlog = cp->hashLog, mls = cp->minMatch, dHBits = cd->cp.hashLog + 8;
    size_t const stepSize = cp->targetLength + !cp->targetLength;
    uint32_t* const hashTable = (uint32_t*)calloc((size_t)1 << hlog, sizeof(uint32_t));
    uint32_t const P = (uint32_t)cd->len + 2;
    const uint8_t* const base = src - P, * const dictBase = cd->content - 2;
    const uint8_t* const dictStart = cd->content, * const dictEnd = cd->content + cd->len;
    const uint8_t* const istart = src, * const iend = src + n, * const ilimit = iend - 8, * const prefixStart = src;
    const uint8_t* ip0 = istart, * ip1 = ip0 + stepSize, * anchor = istart;
    uint32_t offset_1 = rep[0], offset_2 = rep[1];
    while (ip1 <= ilimit) {
        size_t mLength;
        uint32_t hash0 = zo_hash(ip0, hlog, mls);
        uint32_t dHT0 = zo_hash(ip0, dHBits, mls);
        uint32_t dEnt = cd->tabL[dHT0 >> 8];
        int dTags = (dEnt & 0xFF) == (dHT0 & 0xFF);
        uint32_t matchIndex = hashTable[hash0];
        uint32_t curr = (uint32_t)(ip0 - base);
        size_t step = stepSize;
        const uint8_t* nextStep = ip0 + 256;
        for (;;) {
            const uint8_t* match = base + matchIndex;
            uint32_t const repIndex = curr + 1 - offset_1;
            const uint8_t* repMatch = repIndex < P ? dictBase + repIndex : base + repIndex;
            uint32_t const hash1 = zo_hash(ip1, hlog, mls), dHT1 = zo_hash(ip1, dHBits, mls);
            hashTable[hash0] = curr;
            if ((uint32_t)((P - 1) - repIndex) >= 3 && rd32(repMatch) == rd32(ip0 + 1)) {   /* :566 */
                const uint8_t* const repEnd = repIndex < P ? dictEnd : iend;
                mLength = zo_count_2seg(ip0 + 1 + 4, repMatch + 4, iend, repEnd, prefixStart) + 4;
                ip0++;
                SEQ(ip0 - anchor, 1, mLength);
                break;
            }
            if (dTags) {                                                                      /* :575 */
                uint32_t const dIdx = dEnt >> 8;
                const uint8_t* dm = dictBase + dIdx;
                if (dIdx > 2 && rd32(dm) == rd32(ip0) && matchIndex <= P) {
                    uint32_t const offset = curr - dIdx;
                    mLength = zo_count_2seg(ip0 + 4, dm + 4, iend, dictEnd, prefixStart) + 4;
                    while (ip0 > anchor && dm > dictStart && ip0[-1] == dm[-1]) { ip0--; dm--; mLength++; }
                    offset_2 = offset_1; offset_1 = offset;
                    SEQ(ip0 - anchor, offset + 3, mLength);
                    break;
                }
            }
            if (matchIndex >= P && rd32(ip0) == rd32(match)) {                                /* :598 ZSTD_match4Found_cmov */
                uint32_t const offset = (uint32_t)(ip0 - match);
                mLength = zo_count_ptr(ip0 + 4, match + 4, iend) + 4;
                while (ip0 > anchor && match > prefixStart && ip0[-1] == match[-1]) { ip0--; match--; mLength++; }
                offset_2 = offset_1; offset_1 = offset;
                SEQ(ip0 - anchor, offset + 3, mLength);
                break;
            }
            dEnt = cd->tabL[dHT1 >> 8];                                                       /* :614 */
            dTags = (dEnt & 0xFF) == (dHT1 & 0xFF);
            matchIndex = hashTable[hash1];
            if (ip1 >= nextStep) { step++; nextStep += 256; }
            ip0 = ip1; ip1 = ip1 + step;
            if (ip1 > ilimit) goto _cleanup;
            curr = (uint32_t)(ip0 - base);
            hash0 = hash1;
        }
        ip0 += mLength; anchor = ip0;                                                         /* :631 */
        if (ip0 <= ilimit) {
            hashTable[zo_hash(base + curr + 2, hlog, mls)] = curr + 2;
            hashTable[zo_hash(ip0 - 2, hlog, mls)] = (uint32_t)(ip0 - 2 - base);
            while (ip0 <= ilimit) {
                uint32_t const current2 = (uint32_t)(ip0 - base), repIndex2 = current2 - offset_2;
                const uint8_t* repMatch2 = repIndex2 < P ? dictBase + repIndex2 : base + repIndex2;
                if ((uint32_t)((P - 1) - repIndex2) >= 3 && rd32(repMatch2) == rd32(ip0)) {
                    const uint8_t* const repEnd2 = repIndex2 < P ? dictEnd : iend;
                    size_t const rl = zo_count_2seg(ip0 + 4, repMatch2 + 4, iend, repEnd2, prefixStart) + 4;
                    uint32_t const t = offset_2; offset_2 = offset_1; offset_1 = t;
                    SEQ(0, 1, rl);
                    hashTable[zo_hash(ip0, hlog, mls)] = current2;
                    ip0 += rl; anchor = ip0;
                    continue;
                }
                break;
            }
        }
        ip1 = ip0 + stepSize;
    }
_cleanup:
    rep[0] = offset_1; rep[1] = offset_2;
    free(hashTable);
    return (size_t)(iend - anchor);
}
#undef SEQ

/* zstd_fast.c:709-960 ZSTD_compressBlock_fast_extDict_generic — strategy fast in COPY mode (same window layout as
 * zo_dfast_ext below: dictionary = indices 2 .. P-1, source = indices P ..). */
/* src = the start of the source (the prefix segment), the block is src[bStart, bStart + n); Tkeep = the context's table when the blocks
 * of a frame share it (already a copy of the CDict's), NULL = one block with a fresh copy */
static size_t zo_fast_ext_block(const zo_cparams* cp, const zo_cdict* cd, const uint8_t* src, size_t bStart, size_t n, uint32_t* Tkeep, uint32_t dictStartIndex /* window low, >= 2 */, zo_store* st, uint32_t rep[3])
{
    unsigned const hlog = cp->hashLog, mls = cp->minMatch;
    size_t const stepSize = cp->targetLength + !cp->targetLength + 1;
    size_t const sz = (size_t)1 << hlog;
    uint32_t* const T = Tkeep ? Tkeep : (uint32_t*)malloc(sz * sizeof(uint32_t));
    uint32_t const P = (uint32_t)cd->len + 2;
    const uint8_t* const base = src - P, * const dictBase = cd->content - 2;
    const uint8_t* const dictStart = dictBase + dictStartIndex, * const dictEnd = cd->content + cd->len;
    const uint8_t* const istart = src + bStart, * const iend = istart + n, * const ilimit = iend - 8, * const prefixStart = src;
    const uint8_t* anchor = istart, * ip0 = istart, * ip1, * ip2, * ip3, * nextStep, * match0 = NULL, * matchEnd = NULL;
    uint32_t offset_1 = rep[0], offset_2 = rep[1], offsetSaved1 = 0, offsetSaved2 = 0, current0 = 0, idx, offcode = 0;
    uint32_t hash0, hash1;
    size_t step, mLength = 0, i;
#define PTR(k) ((k) < P ? dictBase + (k) : base + (k))
    if (!Tkeep) for (i = 0; i < sz; i++) T[i] = cd->tabL[i] >> 8;                 /* :2379-2393 tags removed */
    {   uint32_t const maxRep = (uint32_t)(ip0 - base) - dictStartIndex;          /* :764-768 */
        if (offset_2 >= maxRep) { offsetSaved2 = offset_2; offset_2 = 0; }
        if (offset_1 >= maxRep) { offsetSaved1 = offset_1; offset_1 = 0; }
    }
    for (;;) {   /* _start */
        int found = 0;
        step = stepSize; nextStep = ip0 + 128;
        ip1 = ip0 + 1; ip2 = ip0 + step; ip3 = ip2 + 1;
        if (ip3 >= ilimit) break;
        hash0 = zo_hash(ip0, hlog, mls); hash1 = zo_hash(ip1, hlog, mls);
        idx = T[hash0];
        do {
            {   uint32_t const current2 = (uint32_t)(ip2 - base), repIndex = current2 - offset_1;             /* :790-817 */
                uint32_t rval;
                if (((uint32_t)(P - repIndex) >= 4) & (offset_1 > 0)) rval = rd32(PTR(repIndex)); else rval = rd32(ip2) ^ 1;
                current0 = (uint32_t)(ip0 - base); T[hash0] = current0;
                if (rd32(ip2) == rval) {
                    ip0 = ip2; match0 = PTR(repIndex); matchEnd = repIndex < P ? dictEnd : iend;
                    mLength = ip0[-1] == match0[-1];
                    ip0 -= mLength; match0 -= mLength;
                    offcode = 1; mLength += 4;
                    found = 2; break;
                }
            }
            if (idx >= dictStartIndex && rd32(PTR(idx)) == rd32(ip0)) { found = 1; break; }                  /* :819-829 */
            idx = T[hash1];                                                                                    /* :831-846 */
            hash0 = hash1; hash1 = zo_hash(ip2, hlog, mls);
            ip0 = ip1; ip1 = ip2; ip2 = ip3;
            current0 = (uint32_t)(ip0 - base); T[hash0] = current0;
            if (idx >= dictStartIndex && rd32(PTR(idx)) == rd32(ip0)) { found = 1; break; }                  /* :848-858 */
            idx = T[hash1];                                                                                    /* :860-880 */
            hash0 = hash1; hash1 = zo_hash(ip2, hlog, mls);
            ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step;
            if (ip2 >= nextStep) { step++; nextStep += 128; }
        } while (ip3 < ilimit);
        if (!found) break;                                                                                     /* _cleanup */
        if (found == 1) {                                                                                      /* _offset :899-915 */
            uint32_t const offset = current0 - idx;
            const uint8_t* const low = idx < P ? dictStart : prefixStart;
            matchEnd = idx < P ? dictEnd : iend;
            match0 = PTR(idx);
            offset_2 = offset_1; offset_1 = offset;
            offcode = offset + 3; mLength = 4;
            while (((ip0 > anchor) & (match0 > low)) && ip0[-1] == match0[-1]) { ip0--; match0--; mLength++; }
        }
        mLength += zo_count_2seg(ip0 + mLength, match0 + mLength, iend, matchEnd, prefixStart);              /* _match :917-957 */
        zo_store_seq(st, src, (size_t)(anchor - src), (size_t)(ip0 - anchor), offcode, (uint32_t)mLength);
        ip0 += mLength; anchor = ip0;
        if (ip1 < ip0) T[hash1] = (uint32_t)(ip1 - base);
        if (ip0 <= ilimit) {
            T[zo_hash(base + current0 + 2, hlog, mls)] = current0 + 2;
            T[zo_hash(ip0 - 2, hlog, mls)] = (uint32_t)(ip0 - 2 - base);
            while (ip0 <= ilimit) {
                uint32_t const repIndex2 = (uint32_t)(ip0 - base) - offset_2;
                const uint8_t* const repMatch2 = PTR(repIndex2);
                if ((((uint32_t)((P - 1) - repIndex2) >= 3) & (offset_2 > 0)) && rd32(repMatch2) == rd32(ip0)) {
                    const uint8_t* const repEnd2 = repIndex2 < P ? dictEnd : iend;
                    size_t const rl = zo_count_2seg(ip0 + 4, repMatch2 + 4, iend, repEnd2, prefixStart) + 4;
                    uint32_t const t = offset_2; offset_2 = offset_1; offset_1 = t;
                    zo_store_seq(st, src, (size_t)(anchor - src), 0, 1, (uint32_t)rl);
                    T[zo_hash(ip0, hlog, mls)] = (uint32_t)(ip0 - base);
                    ip0 += rl; anchor = ip0;
                    continue;
                }
                break;
            }
        }
    }
#undef PTR
    offsetSaved2 = (offsetSaved1 != 0 && offset_1 != 0) ? offsetSaved1 : offsetSaved2;
    rep[0] = offset_1 ? offset_1 : offsetSaved1;
    rep[1] = offset_2 ? offset_2 : offsetSaved2;
    if (!Tkeep) free(T);
    return (size_t)(iend - anchor);
}
static size_t zo_fast_ext(const zo_cparams* cp, const zo_cdict* cd, const uint8_t* src, size_t n, zo_store* st, uint32_t rep[3])
{
    return zo_fast_ext_block(cp, cd, src, 0, n, NULL, 2, st, rep);
}

/* zstd_double_fast.c:551-759 ZSTD_compressBlock_doubleFast_extDict_generic — the COPY mode of a CDict (zstd_compress.c:2395-2470):
 * the working tables start as copies of the CDict's (tags removed, :2379-2393), the dictionary content is the window's
 * extDict segment (indices 2 .. P-1), the source the prefix (indices P ..); one table pair serves both segments. */
#define SEQ(litLen, offBase, ml) zo_store_seq(st, src, (size_t)(anchor - src), (size_t)(litLen), (offBase), (uint32_t)(ml))
#define PTR(idx) ((idx) < P ? dictBase + (idx) : base + (idx))
static size_t zo_dfast_ext_block(const zo_cparams* cp, const zo_cdict* cd, const uint8_t* src, size_t bStart, size_t n, uint32_t* TLkeep, uint32_t* TSkeep, uint32_t dictStartIndex, zo_store* st, uint32_t rep[3])
{
    unsigned const hBitsL = cp->hashLog, hBitsS = cp->chainLog, mls = cp->minMatch;
    size_t const szL = (size_t)1 << hBitsL, szS = (size_t)1 << hBitsS;
    uint32_t* const hashLong = TLkeep ? TLkeep : (uint32_t*)malloc(szL * sizeof(uint32_t));
    uint32_t* const hashSmall = TLkeep ? TSkeep : (uint32_t*)malloc(szS * sizeof(uint32_t));
    uint32_t const P = (uint32_t)cd->len + 2;
    const uint8_t* const base = src - P, * const dictBase = cd->content - 2;
    const uint8_t* const dictStart = dictBase + dictStartIndex, * const dictEnd = cd->content + cd->len;
    const uint8_t* const istart = src + bStart, * const iend = istart + n, * const ilimit = iend - 8, * const prefixStart = src;
    const uint8_t* ip = istart, * anchor = istart;
    uint32_t offset_1 = rep[0], offset_2 = rep[1];
    size_t i;
    if (!TLkeep) {
        for (i = 0; i < szL; i++) hashLong[i] = cd->tabL[i] >> 8;
        for (i = 0; i < szS; i++) hashSmall[i] = cd->tabS[i] >> 8;
    }
    while (ip < ilimit) {
        uint32_t const hSmall = zo_hash(ip, hBitsS, mls), hLong = zo_hash(ip, hBitsL, 8);
        uint32_t const matchIndex = hashSmall[hSmall], matchLongIndex = hashLong[hLong];
        const uint8_t* match = PTR(matchIndex); const uint8_t* matchLong = PTR(matchLongIndex);
        uint32_t const curr = (uint32_t)(ip - base), repIndex = curr + 1 - offset_1;
        const uint8_t* const repMatch = PTR(repIndex);
        size_t mLength;
        hashSmall[hSmall] = hashLong[hLong] = curr;
        if (((uint32_t)((P - 1) - repIndex) >= 3) && (offset_1 <= curr + 1 - dictStartIndex) && rd32(repMatch) == rd32(ip + 1)) {   /* :613 */
            const uint8_t* const repEnd = repIndex < P ? dictEnd : iend;
            mLength = zo_count_2seg(ip + 1 + 4, repMatch + 4, iend, repEnd, prefixStart) + 4;
            ip++;
            SEQ(ip - anchor, 1, mLength);
        } else {
            if (matchLongIndex > dictStartIndex && rd64(matchLong) == rd64(ip)) {             /* :621 */
                const uint8_t* const matchEnd = matchLongIndex < P ? dictEnd : iend;
                const uint8_t* const low = matchLongIndex < P ? dictStart : prefixStart;
                uint32_t offset;
                mLength = zo_count_2seg(ip + 8, matchLong + 8, iend, matchEnd, prefixStart) + 8;
                offset = curr - matchLongIndex;
                while (ip > anchor && matchLong > low && ip[-1] == matchLong[-1]) { ip--; matchLong--; mLength++; }
                offset_2 = offset_1; offset_1 = offset;
                SEQ(ip - anchor, offset + 3, mLength);
            } else if (matchIndex > dictStartIndex && rd32(match) == rd32(ip)) {              /* :633 */
                uint32_t const h3 = zo_hash(ip + 1, hBitsL, 8), matchIndex3 = hashLong[h3];
                const uint8_t* match3 = PTR(matchIndex3);
                uint32_t offset;
                hashLong[h3] = curr + 1;
                if (matchIndex3 > dictStartIndex && rd64(match3) == rd64(ip + 1)) {
                    const uint8_t* const matchEnd = matchIndex3 < P ? dictEnd : iend;
                    const uint8_t* const low = matchIndex3 < P ? dictStart : prefixStart;
                    mLength = zo_count_2seg(ip + 9, match3 + 8, iend, matchEnd, prefixStart) + 8;
                    ip++;
                    offset = curr + 1 - matchIndex3;
                    while (ip > anchor && match3 > low && ip[-1] == match3[-1]) { ip--; match3--; mLength++; }
                } else {
                    const uint8_t* const matchEnd = matchIndex < P ? dictEnd : iend;
                    const uint8_t* const low = matchIndex < P ? dictStart : prefixStart;
                    mLength = zo_count_2seg(ip + 4, match + 4, iend, matchEnd, prefixStart) + 4;
                    offset = curr - matchIndex;
                    while (ip > anchor && match > low && ip[-1] == match[-1]) { ip--; match--; mLength++; }
                }
                offset_2 = offset_1; offset_1 = offset;
                SEQ(ip - anchor, offset + 3, mLength);
            } else { ip += ((ip - anchor) >> 8) + 1; continue; }
        }
        ip += mLength; anchor = ip;
        if (ip <= ilimit) {                                                                   /* :677 */
            uint32_t const ins = curr + 2;
            hashLong[zo_hash(base + ins, hBitsL, 8)] = ins;
            hashLong[zo_hash(ip - 2, hBitsL, 8)] = (uint32_t)(ip - 2 - base);
            hashSmall[zo_hash(base + ins, hBitsS, mls)] = ins;
            hashSmall[zo_hash(ip - 1, hBitsS, mls)] = (uint32_t)(ip - 1 - base);
            while (ip <= ilimit) {
                uint32_t const current2 = (uint32_t)(ip - base), repIndex2 = current2 - offset_2;
                const uint8_t* repMatch2 = PTR(repIndex2);
                if (((uint32_t)((P - 1) - repIndex2) >= 3) && (offset_2 <= current2 - dictStartIndex) && rd32(repMatch2) == rd32(ip)) {
                    const uint8_t* const repEnd2 = repIndex2 < P ? dictEnd : iend;
                    size_t const rl = zo_count_2seg(ip + 4, repMatch2 + 4, iend, repEnd2, prefixStart) + 4;
                    uint32_t const t = offset_2; offset_2 = offset_1; offset_1 = t;
                    SEQ(0, 1, rl);
                    hashSmall[zo_hash(ip, hBitsS, mls)] = current2;
                    hashLong[zo_hash(ip, hBitsL, 8)] = current2;
                    ip += rl; anchor = ip;
                    continue;
                }
                break;
            }
        }
    }
    rep[0] = offset_1; rep[1] = offset_2;
    if (!TLkeep) { free(hashLong); free(hashSmall); }
    return (size_t)(iend - anchor);
}
static size_t zo_dfast_ext(const zo_cparams* cp, const zo_cdict* cd, const uint8_t* src, size_t n, zo_store* st, uint32_t rep[3])
{
    return zo_dfast_ext_block(cp, cd, src, 0, n, NULL, NULL, 2, st, rep);
}
#undef PTR
#undef SEQ

/* working-context parameters for a source of n bytes compressed with `cd` attached, or -1 if the reference would copy the
 * dictionary instead (zstd_compress.c:2289-2315) — only the attach path is restated */
int zo_cdict_params(const zo_cdict* cd, size_t n, zo_cparams* out)
{
    static const size_t cutoff[6] = { 8192, 8192, 16384, 32768, 32768, 32768 };
    zo_cparams p, w;
    if (n > cutoff[cd->cp.strategy]) {
        /* COPY mode (zstd_compress.c:2395-2419): the CDict's table parameters as they are, windowLog from the parameters
         * requested for (level, srcSize, dictSize) with the dictionary counted in (ZSTD_cpm_noAttachDict, :6289-6292).
         * Strategies fast and dfast are restated (ZSTD_compressBlock_{fast,doubleFast}_extDict). */
        if (cd->cp.strategy > 5 || n > ZO_BLOCK_MAX) return -1;
        if (zo_get_cparams_mode(cd->level, n, cd->fullSize, 0, &p) < 0) return -1;
        w = cd->cp; w.windowLog = p.windowLog;
        *out = w;
        return 1;
    }
    {   int r; g_zo_any_strategy = 1; r = zo_get_cparams_mode(cd->level, n, cd->len, 1, &p); g_zo_any_strategy = 0;   /* only its windowLog is used: any strategy's row will do */
        if (r < 0) return -1; }                                                  /* :6289-6292 requested params, attach mode */
    w = cd->cp;                                                                  /* :2331-2335 */
    zo_adjust_cparams(&w, n, cd->len, 1);
    w.windowLog = p.windowLog;
    *out = w;
    return 0;
}

/* ------------------------------------------------------------------ block + frame */
/* zstd_compress.c:4626-4672 with defaults (content size on, no checksum, no dictID) */
static size_t write_frame_header_dict(uint8_t* op, const zo_cparams* cp, unsigned long long n, uint32_t dictID)
{
    uint32_t const windowSize = 1u << cp->windowLog;
    unsigned const single = windowSize >= n;
    unsigned const fcs = (n >= 256) + (n >= 65536 + 256) + (n >= 0xFFFFFFFFU);
    unsigned const dcode = (dictID > 0) + (dictID >= 256) + (dictID >= 65536);   /* :4634 dictIDSizeCode */
    size_t pos = 4;
    wr32(op, 0xFD2FB528U);
    op[pos++] = (uint8_t)(dcode + (single << 5) + (fcs << 6));
    if (!single) op[pos++] = (uint8_t)((cp->windowLog - 10) << 3);
    switch (dcode) {                                                             /* :4651-4657 */
    case 1: op[pos++] = (uint8_t)dictID; break;
    case 2: wr16(op + pos, dictID); pos += 2; break;
    case 3: wr32(op + pos, dictID); pos += 4; break;
    default: break;
    }
    switch (fcs) {
    case 0: if (single) op[pos++] = (uint8_t)n; break;
    case 1: wr16(op + pos, (unsigned)(n - 256)); pos += 2; break;
    case 2: wr32(op + pos, (uint32_t)n); pos += 4; break;
    default: wr32(op + pos, (uint32_t)n); wr32(op + pos + 4, (uint32_t)(n >> 32)); pos += 8; break;
    }
    return pos;
}

static size_t write_frame_header(uint8_t* op, const zo_cparams* cp, unsigned long long n) { return write_frame_header_dict(op, cp, n, 0); }

size_t zo_compress_unit_params(void* dstv, size_t cap, const void* srcv, size_t n, const zo_cparams* cp)
{
    uint8_t* const dst = (uint8_t*)dstv; const uint8_t* const src = (const uint8_t*)srcv;
    uint8_t* op = dst; size_t cSize = 0;
    if (n > ZO_BLOCK_MAX || cap < zo_compress_bound(n)) return ZO_ERROR;
    op += write_frame_header(op, cp, n);
    if (n == 0) { wr24(op, 1); return (size_t)(op + 3 - dst); }                  /* zstd_compress.c:5270: empty last raw block */
    if (n >= 7) {                                                                 /* :3216 MIN_CBLOCK_SIZE+3+1+1 */
        zo_seq* seqs = (zo_seq*)malloc(sizeof(zo_seq) * (n / 3 + 2));
        uint8_t* lits = (uint8_t*)malloc(n + 8);
        size_t litSize = 0;
        size_t const nb = zo_parse_block(cp, src, n, seqs, n / 3 + 2, lits, &litSize, NULL);
        uint8_t* body = op + 3;
        if (nb == ZO_ERROR) { free(seqs); free(lits); return ZO_ERROR; }
        {   int const suspect = (nb == 0) || (litSize / nb >= 20);               /* :2918 */
            size_t const l = zo_compress_literals(body, cap, lits, litSize, cp, suspect);
            size_t const s = zo_compress_sequences(body + l, cap, seqs, nb, cp);
            if (s == ZO_ERROR) { free(seqs); free(lits); return ZO_ERROR; }
            cSize = (s == 0) ? 0 : l + s;
            if (cSize >= n - ((n >> 6) + 2)) cSize = 0;                           /* :3026 minGain */
        }
        free(seqs); free(lits);
    }
    if (cSize == 0) { wr24(op, (uint32_t)(1 + (0 << 1) + (n << 3))); memcpy(op + 3, src, n); return (size_t)(op + 3 + n - dst); }
    wr24(op, (uint32_t)(1 + (2 << 1) + (cSize << 3)));                           /* :4586-4590 (first block: never RLE) */
    return (size_t)(op + 3 + cSize - dst);
}

/* debugging aid for the tests: (litLength, matchLength, offBase) of one source parsed with `cd` attached */
size_t zo_parse_cdict(const zo_cdict* cd, const void* srcv, size_t n, uint32_t* out, size_t capSeqs)
{
    const uint8_t* const src = (const uint8_t*)srcv;
    zo_cparams cp; zo_store st; uint32_t rep[3]; size_t i;
    zo_seq* seqs = (zo_seq*)malloc(sizeof(zo_seq) * (n / 3 + 2));
    uint8_t* lits = (uint8_t*)malloc(n + 8);
    int const mode = zo_cdict_params(cd, n, &cp);
    if (mode < 0 || n < 8) { free(seqs); free(lits); return ZO_ERROR; }
    rep[0] = cd->rep[0]; rep[1] = cd->rep[1]; rep[2] = cd->rep[2];
    st.seqs = seqs; st.nb = 0; st.cap = n / 3 + 2; st.lits = lits; st.litSize = 0; st.overflow = 0;
    if (mode == 1) { if (cp.strategy == 1) zo_fast_ext(&cp, cd, src, n, &st, rep); else zo_dfast_ext(&cp, cd, src, n, &st, rep); }
    else if (cp.strategy == 1) zo_fast_dms(&cp, cd, src, n, &st, rep); else zo_dfast_dms(&cp, cd, src, n, &st, rep);
    for (i = 0; i < st.nb && i < capSeqs; i++) { out[3*i] = seqs[i].litLength; out[3*i+1] = seqs[i].matchLength; out[3*i+2] = seqs[i].offBase; }
    free(seqs); free(lits);
    return st.nb;
}

/* ZSTD_compress2 with ZSTD_CCtx_refCDict(cd) on one source of n bytes (n <= the attach cutoff): one frame, one block */
size_t zo_compress_unit_cdict(void* dstv, size_t cap, const void* srcv, size_t n, const zo_cdict* cd)
{
    uint8_t* const dst = (uint8_t*)dstv; const uint8_t* const src = (const uint8_t*)srcv;
    uint8_t* op = dst; size_t cSize = 0;
    zo_cparams cp;
    int const mode = zo_cdict_params(cd, n, &cp);                                /* 0 attach, 1 copy */
    if (mode < 0 || cap < zo_compress_bound(n)) return ZO_ERROR;
    if (cd->len == 0) return zo_compress_unit_params(dstv, cap, srcv, n, &cp);  /* :2354 an empty dictionary is not attached, its parameters still apply */
    op += write_frame_header_dict(op, &cp, n, cd->dictID);
    if (n == 0) { wr24(op, 1); return (size_t)(op + 3 - dst); }
    if (n >= 7) {
        zo_seq* seqs = (zo_seq*)malloc(sizeof(zo_seq) * (n / 3 + 2));
        uint8_t* lits = (uint8_t*)malloc(n + 8);
        zo_store st; uint32_t rep[3]; size_t last;
        uint8_t* body = op + 3;
        rep[0] = cd->rep[0]; rep[1] = cd->rep[1]; rep[2] = cd->rep[2];
        st.seqs = seqs; st.nb = 0; st.cap = n / 3 + 2; st.lits = lits; st.litSize = 0; st.overflow = 0;
        if (n < 8) last = n;
        else if (mode == 1) last = cp.strategy >= 3 ? zo_lazy_ext(&cp, cd, src, n, &st, rep, cp.strategy - 3)
                                : cp.strategy == 1 ? zo_fast_ext(&cp, cd, src, n, &st, rep) : zo_dfast_ext(&cp, cd, src, n, &st, rep);
        else if (cp.strategy >= 3) last = zo_lazy_dms(&cp, cd, src, n, &st, rep, cp.strategy - 3);
        else if (cp.strategy == 1) last = zo_fast_dms(&cp, cd, src, n, &st, rep);
        else last = zo_dfast_dms(&cp, cd, src, n, &st, rep);
        memcpy(lits + st.litSize, src + n - last, last);
        st.litSize += last;
        {   int const suspect = (st.nb == 0) || (st.litSize / st.nb >= 20);
            const zo_prev* const pv = cd->hasEntropy ? &cd->prev : NULL;
            size_t const l = zo_compress_literals_prev(body, cap, lits, st.litSize, &cp, suspect, pv);
            size_t const q = zo_compress_sequences_prev(body + l, cap, seqs, st.nb, &cp, pv);
            if (q == ZO_ERROR || st.overflow) { free(seqs); free(lits); return ZO_ERROR; }
            cSize = (q == 0) ? 0 : l + q;
            if (cSize >= n - ((n >> 6) + 2)) cSize = 0;
        }
        free(seqs); free(lits);
    }
    if (cSize == 0) { wr24(op, (uint32_t)(1 + (0 << 1) + (n << 3))); memcpy(op + 3, src, n); return (size_t)(op + 3 + n - dst); }
    wr24(op, (uint32_t)(1 + (2 << 1) + (cSize << 3)));
    return (size_t)(op + 3 + cSize - dst);
}

/* XXH64 (lib/common/xxhash.h, the public-domain algorithm of the xxHash specification, seed = 0 for zstd frames) */
static uint64_t xxh_rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint64_t xxh_round(uint64_t acc, uint64_t in) { acc += in * 0xC2B2AE3D27D4EB4FULL; return xxh_rotl(acc, 31) * 0x9E3779B185EBCA87ULL; }
uint64_t zo_xxh64(const void* srcv, size_t n, uint64_t seed)
{
    uint64_t const P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
    const uint8_t* p = (const uint8_t*)srcv; const uint8_t* const end = p + n;
    uint64_t h;
    if (n >= 32) {
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        do { v1 = xxh_round(v1, rd64(p)); v2 = xxh_round(v2, rd64(p + 8)); v3 = xxh_round(v3, rd64(p + 16)); v4 = xxh_round(v4, rd64(p + 24)); p += 32; } while (p + 32 <= end);
        h = xxh_rotl(v1, 1) + xxh_rotl(v2, 7) + xxh_rotl(v3, 12) + xxh_rotl(v4, 18);
        h = (h ^ xxh_round(0, v1)) * P1 + P4; h = (h ^ xxh_round(0, v2)) * P1 + P4;
        h = (h ^ xxh_round(0, v3)) * P1 + P4; h = (h ^ xxh_round(0, v4)) * P1 + P4;
    } else h = seed + P5;
    h += (uint64_t)n;
    while (p + 8 <= end) { h ^= xxh_round(0, rd64(p)); h = xxh_rotl(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P1; h = xxh_rotl(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (uint64_t)(*p++) * P5; h = xxh_rotl(h, 11) * P1; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

/* ZSTD_c_checksumFlag = 1 (zstd_compress.c:4637, :5297-5303): descriptor bit 2 + LE32 of the low half of XXH64(content) after
 * the last block; nothing else in the frame changes */
size_t zo_frame_add_checksum(void* framev, size_t frameSize, const void* src, size_t n)
{
    uint8_t* const f = (uint8_t*)framev;
    f[4] |= 1u << 2;
    wr32(f + frameSize, (uint32_t)zo_xxh64(src, n, 0));
    return frameSize + 4;
}

/* ------------------------------------------------------------------ multi-block frames
 * ZSTD_compress2 of a source of any size as ONE frame: ZSTD_compress_frameChunk (zstd_compress.c:4527-4623) with the blind
 * 92 KB split (:4494-4518), blocks sharing window, hash table, repcodes and the previous Huffman table
 * (ZSTD_blockState_confirmRepcodesAndEntropyTables :3549, only after a block that was emitted compressed :4379-4381).
 * Strategy ZSTD_fast (zstd_fast.c:192-423 with a prefix): positions are relative to the frame start; T[] holds pos+1. */
/* lowest position of the window the block's context has seen: 0 for a frame compressed in one piece, the start of the overlap
 * prefix for a job of the multi-threaded frame (zstdmt_compress.c) */
static size_t g_zo_win_start = 0;

static size_t zo_fast_block(const zo_cparams* cp, const uint8_t* src /* frame start */, size_t bStart, size_t bLen, uint32_t* T,
                            zo_store* st, uint32_t rep[3])
{
    unsigned const hlog = cp->hashLog, mls = cp->minMatch;
    size_t const stepSize = cp->targetLength + !cp->targetLength + 1;
    size_t const maxDist = (size_t)1 << cp->windowLog;
    size_t const iend = bStart + bLen, ilimit = iend - 8;
    size_t const dl0 = bStart > maxDist ? bStart - maxDist : 0;                  /* ZSTD_window_enforceMaxDist (:1107) from the block START */
    size_t const dictLimit = dl0 > g_zo_win_start ? dl0 : g_zo_win_start;
    size_t const prefixLow = (iend - dictLimit > maxDist) ? iend - maxDist : dictLimit;   /* ZSTD_getLowestPrefixIndex(endIndex) :1206 */
    size_t anchor = bStart, ip0 = bStart, ip1, ip2, ip3, cur0 = 0, step, nextStep, match0 = 0, mLength;
    uint32_t rep1 = rep[0], rep2 = rep[1], saved1 = 0, saved2 = 0, offBase;
    uint32_t h0, h1, cand;
    if (iend < 8) return bLen;                                                   /* a 7-byte frame: ilimit lies before the source (the reference compares pointers), nothing is searched */
    ip0 += (ip0 == prefixLow);                                                   /* :238 */
    {   size_t const windowLow = (ip0 - dictLimit > maxDist) ? ip0 - maxDist : dictLimit;   /* :239-244 */
        size_t const maxRep = ip0 - windowLow;
        if (rep2 > maxRep) { saved2 = rep2; rep2 = 0; }
        if (rep1 > maxRep) { saved1 = rep1; rep1 = 0; }
    }
#define ZO_VALID(c) ((c) != 0 && (size_t)(c) - 1 >= prefixLow)
    for (;;) {
        step = stepSize; nextStep = ip0 + 128;
        ip1 = ip0 + 1; ip2 = ip0 + step; ip3 = ip2 + 1;
        if (ip3 >= ilimit) break;
        h0 = zo_hash(src + ip0, hlog, mls); h1 = zo_hash(src + ip1, hlog, mls);
        cand = T[h0];
        for (;;) {
            int found = 0;
            uint32_t const rval = rep1 ? rd32(src + ip2 - rep1) : 0;
            cur0 = ip0; T[h0] = (uint32_t)ip0 + 1;
            if (rep1 > 0 && rd32(src + ip2) == rval) {
                ip0 = ip2; match0 = ip0 - rep1;
                mLength = (src[ip0 - 1] == src[match0 - 1]);
                ip0 -= mLength; match0 -= mLength;
                offBase = 1; mLength += 4;
                T[h1] = (uint32_t)ip1 + 1;
                found = 2;
            } else if (ZO_VALID(cand) && rd32(src + ip0) == rd32(src + cand - 1)) {
                T[h1] = (uint32_t)ip1 + 1;
                found = 1;
            } else {
                cand = T[h1]; h0 = h1; h1 = zo_hash(src + ip2, hlog, mls);
                ip0 = ip1; ip1 = ip2; ip2 = ip3;
                cur0 = ip0; T[h0] = (uint32_t)ip0 + 1;
                if (ZO_VALID(cand) && rd32(src + ip0) == rd32(src + cand - 1)) {
                    if (step <= 4) T[h1] = (uint32_t)ip1 + 1;
                    found = 1;
                } else {
                    cand = T[h1]; h0 = h1; h1 = zo_hash(src + ip2, hlog, mls);
                    ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step;
                    if (ip2 >= nextStep) { step++; nextStep += 128; }
                    if (ip3 < ilimit) continue;
                    goto cleanup;
                }
            }
            if (found == 1) {
                match0 = cand - 1;
                rep2 = rep1; rep1 = (uint32_t)(ip0 - match0);
                offBase = rep1 + 3; mLength = 4;
                while (ip0 > anchor && match0 > prefixLow && src[ip0 - 1] == src[match0 - 1]) { ip0--; match0--; mLength++; }
            }
            {   size_t a = ip0 + mLength, b = match0 + mLength;                  /* ZSTD_count up to the BLOCK end */
                while (a < iend && src[a] == src[b]) { a++; b++; }
                mLength = a - ip0;
            }
            zo_store_seq(st, src, anchor, ip0 - anchor, offBase, (uint32_t)mLength);
            ip0 += mLength; anchor = ip0;
            if (ip0 <= ilimit) {
                T[zo_hash(src + cur0 + 2, hlog, mls)] = (uint32_t)cur0 + 2 + 1;
                T[zo_hash(src + ip0 - 2, hlog, mls)] = (uint32_t)ip0 - 2 + 1;
                if (rep2 > 0) {
                    while (ip0 <= ilimit && rd32(src + ip0) == rd32(src + ip0 - rep2)) {
                        size_t a = ip0 + 4, b = ip0 + 4 - rep2;
                        uint32_t rLength;
                        while (a < iend && src[a] == src[b]) { a++; b++; }
                        rLength = (uint32_t)(a - ip0);
                        {   uint32_t const t = rep2; rep2 = rep1; rep1 = t; }
                        T[zo_hash(src + ip0, hlog, mls)] = (uint32_t)ip0 + 1;
                        ip0 += rLength;
                        zo_store_seq(st, src, anchor, 0, 1, rLength);
                        anchor = ip0;
                    }
                }
            }
            break;
        }
    }
cleanup:
#undef ZO_VALID
    saved2 = (saved1 != 0 && rep1 != 0) ? saved1 : saved2;
    rep[0] = rep1 ? rep1 : saved1;
    rep[1] = rep2 ? rep2 : saved2;
    return iend - anchor;
}

/* zstd_double_fast.c:105-323 for one block of a multi-block frame: the tables carry over from the previous blocks, matches may
 * start anywhere in [prefixLow, position).  Table entries are pos+1 (0 = empty), positions relative to the frame start.
 * Validity of a candidate, in the reference's own asymmetric terms: long / short match at ip: index >= prefixLowestIndex
 * (ZSTD_selectAddr, :200, :214); long match at ip+1: index > prefixLowestIndex (:260); catch-up: match > prefixLowest (:207, :267). */
static size_t zo_dfast_block(const zo_cparams* cp, const uint8_t* src /* frame start */, size_t bStart, size_t bLen, uint32_t* TL, uint32_t* TS,
                             zo_store* st, uint32_t rep[3])
{
    unsigned const hL = cp->hashLog, hS = cp->chainLog, mls = cp->minMatch;
    size_t const maxDist = (size_t)1 << cp->windowLog;
    size_t const iend = bStart + bLen, ilimit = iend - 8;
    size_t const dl0 = bStart > maxDist ? bStart - maxDist : 0;
    size_t const dictLimit = dl0 > g_zo_win_start ? dl0 : g_zo_win_start;
    size_t const prefixLow = (iend - dictLimit > maxDist) ? iend - maxDist : dictLimit;
    size_t anchor = bStart, ip = bStart, ip1, step, nextStep, curr = 0, mLength = 0;
    uint32_t off1 = rep[0], off2 = rep[1], saved1 = 0, saved2 = 0, offset = 0;
    if (iend < 8) return bLen;                                                   /* a 7-byte frame, as in zo_fast_block */
    ip += (ip == prefixLow);                                                     /* :157 */
    {   size_t const windowLow = (ip - dictLimit > maxDist) ? ip - maxDist : dictLimit;
        size_t const maxRep = ip - windowLow;
        if (off2 > maxRep) { saved2 = off2; off2 = 0; }
        if (off1 > maxRep) { saved1 = off1; off1 = 0; }
    }
#define ZO_GE(idx) ((idx) != 0 && (size_t)(idx) - 1 >= prefixLow)
#define ZO_GT(idx) ((idx) != 0 && (size_t)(idx) - 1 > prefixLow)
    for (;;) {
        uint32_t hl0, hl1 = 0, idxl0, idxl1 = 0;
        int kind = 0;
        step = 1; nextStep = ip + 256; ip1 = ip + step;
        if (ip1 > ilimit) break;
        hl0 = zo_hash(src + ip, hL, 8); idxl0 = TL[hl0];
        do {
            uint32_t const hs0 = zo_hash(src + ip, hS, mls);
            uint32_t const idxs0 = TS[hs0];
            size_t matchs0;
            curr = ip;
            TL[hl0] = TS[hs0] = (uint32_t)ip + 1;
            if (off1 > 0 && rd32(src + ip + 1 - off1) == rd32(src + ip + 1)) {
                mLength = zo_count(src, ip + 1 + 4, ip + 1 + 4 - off1, iend) + 4;
                ip++;
                zo_store_seq(st, src, anchor, ip - anchor, 1, (uint32_t)mLength);
                kind = 1; break;
            }
            hl1 = zo_hash(src + ip1, hL, 8);
            if (ZO_GE(idxl0) && rd64(src + idxl0 - 1) == rd64(src + ip)) {
                size_t m = idxl0 - 1;
                mLength = zo_count(src, ip + 8, m + 8, iend) + 8;
                offset = (uint32_t)(ip - m);
                while (ip > anchor && m > prefixLow && src[ip - 1] == src[m - 1]) { ip--; m--; mLength++; }
                kind = 2; break;
            }
            idxl1 = TL[hl1];
            if (ZO_GE(idxs0) && rd32(src + idxs0 - 1) == rd32(src + ip)) {
                matchs0 = idxs0 - 1;
                mLength = zo_count(src, ip + 4, matchs0 + 4, iend) + 4;
                offset = (uint32_t)(ip - matchs0);
                if (ZO_GT(idxl1) && rd64(src + idxl1 - 1) == rd64(src + ip1)) {
                    size_t const m1 = idxl1 - 1;
                    size_t const l1len = zo_count(src, ip1 + 8, m1 + 8, iend) + 8;
                    if (l1len > mLength) { ip = ip1; mLength = l1len; offset = (uint32_t)(ip - m1); matchs0 = m1; }
                }
                while (ip > anchor && matchs0 > prefixLow && src[ip - 1] == src[matchs0 - 1]) { ip--; matchs0--; mLength++; }
                kind = 2; break;
            }
            if (ip1 >= nextStep) { step++; nextStep += 256; }
            ip = ip1; ip1 += step;
            hl0 = hl1; idxl0 = idxl1;
        } while (ip1 <= ilimit);
        if (!kind) break;
        if (kind == 2) {
            off2 = off1; off1 = offset;
            if (step < 4) TL[hl1] = (uint32_t)ip1 + 1;
            zo_store_seq(st, src, anchor, ip - anchor, offset + 3, (uint32_t)mLength);
        }
        ip += mLength; anchor = ip;
        if (ip <= ilimit) {
            size_t const ins = curr + 2;
            TL[zo_hash(src + ins, hL, 8)] = (uint32_t)ins + 1;
            TL[zo_hash(src + ip - 2, hL, 8)] = (uint32_t)ip - 2 + 1;
            TS[zo_hash(src + ins, hS, mls)] = (uint32_t)ins + 1;
            TS[zo_hash(src + ip - 1, hS, mls)] = (uint32_t)ip - 1 + 1;
            while (ip <= ilimit && off2 > 0 && rd32(src + ip) == rd32(src + ip - off2)) {
                uint32_t const rLength = zo_count(src, ip + 4, ip + 4 - off2, iend) + 4;
                uint32_t const t = off2; off2 = off1; off1 = t;
                TS[zo_hash(src + ip, hS, mls)] = (uint32_t)ip + 1;
                TL[zo_hash(src + ip, hL, 8)] = (uint32_t)ip + 1;
                zo_store_seq(st, src, anchor, 0, 1, rLength);
                ip += rLength; anchor = ip;
            }
        }
    }
#undef ZO_GE
#undef ZO_GT
    saved2 = (saved1 != 0 && off1 != 0) ? saved1 : saved2;
    rep[0] = off1 ? off1 : saved1;
    rep[1] = off2 ? off2 : saved2;
    return iend - anchor;
}

size_t zo_frame_bound(size_t n) { return n + (n >> 8) + 64 + 3 * (n / 8192 + 2); }

/* the state one compression context carries from block to block (ZSTD_CCtx: match state + blockState.prevCBlock) */
typedef struct {
    uint32_t* T;              /* fast: one table; dfast: long table followed by the short table */
    zo_prev prev;             /* previous block's Huffman table */
    uint32_t rep[3];
    int isFirst;              /* zc->isFirstBlock */
    long long savings;        /* consumedSrcSize - producedCSize of the context (zstd_compress.c:4538): what frame chunks inherit from one another */
    zo_seq* seqs; uint8_t* lits; uint8_t* body;
    zo_lz lz;                 /* greedy / lazy / lazy2: hash chain or rows, nextToUpdate, window low */
    const zo_cdict* cd;       /* a dictionary in copy mode: T starts as a copy of its tables, its content is the extDict segment of every block */
    int dictValid;            /* ms->loadedDictEnd != 0: the whole dictionary is still inside the window */
    uint32_t winLowIdx;       /* window.lowLimit as a reference index (dictionary byte j = j + 2, source byte i = dictLen + 2 + i) */
} zo_fctx;

/* zstd_preSplit.c:139-181 ZSTD_splitBlock(split_lvl1) = ZSTD_splitBlock_byChunks with one 2-byte event out of five: what ZSTD_lazy2
 * uses to place a block boundary inside the next 128 KB once the frame has saved 3 bytes (zstd_compress.c:4510-4511).  8 KB chunks;
 * the fingerprint of what came before against the next chunk's; the first "too different" chunk starts the next block. */
typedef struct { unsigned events[1024]; size_t nbEvents; } zo_fp;
static void zo_fp_record(zo_fp* fp, const uint8_t* p, size_t n)                  /* :47-58, :79-83 */
{
    size_t const limit = n - 2 + 1; size_t i;
    memset(fp, 0, sizeof(*fp));
    for (i = 0; i < limit; i += 5) fp->events[(uint32_t)(rd16(p + i) * 0x9e3779b9u) >> (32 - 10)]++;
    fp->nbEvents += limit / 5;
}
static size_t zo_split_block_lvl1(const uint8_t* p, size_t srcSize)
{
    static zo_fp past, cur;
    int penalty = 3; size_t pos;
    if (srcSize <= ZO_BLOCK_MAX) return srcSize;
    zo_fp_record(&past, p, 8192);
    for (pos = 8192; pos <= ZO_BLOCK_MAX - 8192; pos += 8192) {
        uint64_t deviation = 0, threshold; size_t k;
        zo_fp_record(&cur, p + pos, 8192);
        for (k = 0; k < 1024; k++) {                                             /* :87-97 fpDistance */
            int64_t const d = (int64_t)past.events[k] * (int64_t)cur.nbEvents - (int64_t)cur.events[k] * (int64_t)past.nbEvents;
            deviation += (uint64_t)(d < 0 ? -d : d);
        }
        threshold = (uint64_t)past.nbEvents * (uint64_t)cur.nbEvents * (uint64_t)(14 + penalty) / 16;   /* :102-114 */
        if (deviation >= threshold) return pos;
        for (k = 0; k < 1024; k++) past.events[k] += cur.events[k];
        past.nbEvents += cur.nbEvents;
        if (penalty > 0) penalty--;
    }
    return ZO_BLOCK_MAX;
}

/* ZSTD_getLowestMatchIndex at the block END for a context whose dictionary is an extDict segment (zstd_compress_internal.h:1312) */
static uint32_t lowLimit_of(const zo_fctx* f, const zo_cparams* cp, size_t pos, size_t bLen)
{
    uint32_t const P = (uint32_t)f->cd->len + 2; size_t const maxDist = (size_t)1 << cp->windowLog;
    size_t const endIndex = (size_t)P + pos + bLen;
    return f->dictValid ? f->winLowIdx : (endIndex - f->winLowIdx > maxDist ? (uint32_t)(endIndex - maxDist) : f->winLowIdx);
}

/* ZSTD_compress_frameChunk (zstd_compress.c:4527-4623) over src[pos, pos + len): `savings` starts from the context's running total; lastChunk = the
 * call that ends the frame (its final block carries the last-block bit).  Returns the bytes written at op, ZO_ERROR on failure. */
static size_t zo_frame_chunk(zo_fctx* f, const zo_cparams* cp, const uint8_t* src, size_t pos, size_t len, int lastChunk, uint8_t* op0)
{
    uint8_t* op = op0;
    size_t const end = pos + len;
    long long savings = f->savings;
    while (pos < end) {
        size_t const remaining = end - pos;
        size_t bLen = remaining < ZO_BLOCK_MAX ? remaining : ZO_BLOCK_MAX;       /* :4494-4518 ZSTD_optimalBlockSize, strategies below lazy2 */
        size_t cSize = 0; int last;
        if (remaining >= ZO_BLOCK_MAX && savings >= 3) bLen = cp->strategy >= 5 ? zo_split_block_lvl1(src + pos, remaining) : 92 * 1024;
        last = lastChunk && bLen == remaining;
        if (bLen >= 7) {                                                         /* :3216 */
            zo_store st; uint32_t nrep[3] = { f->rep[0], f->rep[1], f->rep[2] };
            size_t lastLits; zo_prev next;
            st.seqs = f->seqs; st.nb = 0; st.cap = ZO_BLOCK_MAX / 3 + 2; st.lits = f->lits; st.litSize = 0; st.overflow = 0;
            if (f->cd) {
                /* ZSTD_checkDictValidity (block end) and ZSTD_window_enforceMaxDist (block start), zstd_compress.c:4553-4556; then which
                 * parser: the extDict one while part of the dictionary is inside the window — its own bound is taken at the block END
                 * (zstd_fast.c:722-735 / zstd_double_fast.c:571-590) and it hands over to the plain parser when that bound passes the
                 * dictionary; from then on the tables hold source positions only (entries of the dictionary are dropped) */
                uint32_t const P = (uint32_t)f->cd->len + 2; size_t const maxDist = (size_t)1 << cp->windowLog;
                uint32_t lowLimit;
                if (f->dictValid && pos + bLen > maxDist) f->dictValid = 0;
                if ((size_t)P + pos > maxDist + (f->dictValid ? P : 0)) {
                    uint32_t const newLow = (uint32_t)(P + pos - maxDist);
                    if (f->winLowIdx < newLow) f->winLowIdx = newLow;
                    f->dictValid = 0;
                }
                {   size_t const endIndex = (size_t)P + pos + bLen;
                    lowLimit = f->dictValid ? f->winLowIdx : (endIndex - f->winLowIdx > maxDist ? (uint32_t)(endIndex - maxDist) : f->winLowIdx); }
                if (f->winLowIdx >= P || lowLimit >= P) {
                    size_t i, words = ((size_t)1 << cp->hashLog) + (cp->strategy == 2 ? (size_t)1 << cp->chainLog : 0);
                    for (i = 0; i < words; i++) f->T[i] = f->T[i] >= P ? f->T[i] - P + 1 : 0;      /* reference index -> position + 1, 0 = empty */
                    f->cd = NULL;
                }
            }
            lastLits = f->cd ? (cp->strategy == 2 ? zo_dfast_ext_block(cp, f->cd, src, pos, bLen, f->T, f->T + ((size_t)1 << cp->hashLog), lowLimit_of(f, cp, pos, bLen), &st, nrep)
                                                  : zo_fast_ext_block(cp, f->cd, src, pos, bLen, f->T, lowLimit_of(f, cp, pos, bLen), &st, nrep))
                     : cp->strategy >= 3 ? zo_lazy_block(cp, src, pos, bLen, &f->lz, &st, nrep, cp->strategy - 3)
                     : cp->strategy == 2 ? zo_dfast_block(cp, src, pos, bLen, f->T, f->T + ((size_t)1 << cp->hashLog), &st, nrep)
                                         : zo_fast_block(cp, src, pos, bLen, f->T, &st, nrep);
            memcpy(f->lits + st.litSize, src + pos + bLen - lastLits, lastLits); st.litSize += lastLits;
            next = f->prev;
            {   int const suspect = (st.nb == 0) || (st.litSize / st.nb >= 20);
                size_t l, sq;
                g_huf_next = &next;
                l = zo_compress_literals_prev(f->body, ZO_BLOCK_MAX + 1024, f->lits, st.litSize, cp, suspect, &f->prev);
                g_huf_next = NULL;
                if (l >= 1 && (f->body[0] & 3) < 2) next = f->prev;              /* raw / RLE literals: the previous table stays (literals.c:186, :199) */
                g_fse_next = &next;                                              /* the FSE tables and their repeat states carry over like the Huffman table */
                sq = zo_compress_sequences_prev(f->body + l, ZO_BLOCK_MAX + 1024 - l, f->seqs, st.nb, cp, &f->prev);
                g_fse_next = NULL;
                if (sq == ZO_ERROR || st.overflow) return ZO_ERROR;
                cSize = (sq == 0) ? 0 : l + sq;
                if (cSize >= bLen - ((bLen >> 6) + 2)) cSize = 0;                /* :3026 */
            }
            if (!f->isFirst && cSize < 25) {                                     /* :4365-4376 an RLE block, never the context's first */
                size_t i; int same = 1;
                for (i = 1; i < bLen; i++) if (src[pos + i] != src[pos]) { same = 0; break; }
                if (same) { cSize = 1; f->body[0] = src[pos]; }
            }
            if (cSize > 1) { f->rep[0] = nrep[0]; f->rep[1] = nrep[1]; f->rep[2] = nrep[2]; f->prev = next; }   /* :4379-4381 */
            if (f->prev.ofRepeat == 2) f->prev.ofRepeat = 1;                     /* :3365-3367 offset codes of a dictionary are only trusted for the first block */
        }
        if (cSize == 0) { wr24(op, (uint32_t)(last + (0 << 1) + (bLen << 3))); memcpy(op + 3, src + pos, bLen); cSize = 3 + bLen; }
        else if (cSize == 1) { wr24(op, (uint32_t)(last + (1 << 1) + (bLen << 3))); op[3] = f->body[0]; cSize = 4; }
        else { wr24(op, (uint32_t)(last + (2 << 1) + (cSize << 3))); memcpy(op + 3, f->body, cSize); cSize += 3; }
        op += cSize;
        savings += (long long)bLen - (long long)cSize;
        pos += bLen; f->isFirst = 0;
    }
    f->savings = savings;
    return (size_t)(op - op0);
}

static int zo_fctx_init(zo_fctx* f, const zo_cparams* cp)
{
    memset(f, 0, sizeof(*f));
    if (cp->strategy >= 3) { if (!zo_lz_init(&f->lz, cp)) return 0; f->T = (uint32_t*)calloc(1, sizeof(uint32_t)); }
    else
    f->T = (uint32_t*)calloc(((size_t)1 << cp->hashLog) + (cp->strategy == 2 ? (size_t)1 << cp->chainLog : 0), sizeof(uint32_t));   /* dfast: long table, then short table */
    f->seqs = (zo_seq*)malloc(sizeof(zo_seq) * (ZO_BLOCK_MAX / 3 + 2));
    f->lits = (uint8_t*)malloc(ZO_BLOCK_MAX + 8);
    f->body = (uint8_t*)malloc(ZO_BLOCK_MAX + 1024);
    f->isFirst = 1;
    return f->T && f->seqs && f->lits && f->body;
}
static void zo_fctx_free(zo_fctx* f) { free(f->T); free(f->seqs); free(f->lits); free(f->body); zo_lz_free(&f->lz); }

size_t zo_compress_frame_params(void* dstv, size_t cap, const void* srcv, size_t n, const zo_cparams* cp)
{
    uint8_t* const dst = (uint8_t*)dstv; const uint8_t* const src = (const uint8_t*)srcv;
    uint8_t* op = dst;
    zo_fctx f; size_t r;
    if (cp->strategy < 1 || cp->strategy > 5 || cap < zo_frame_bound(n)) return ZO_ERROR;   /* fast .. lazy2 */
    if (cp->windowLog < 17 && n > ((size_t)1 << cp->windowLog)) return ZO_ERROR;   /* the block size would follow the window (zstd_compress.c: blockSizeMax): not restated */
    op += write_frame_header(op, cp, n);
    if (n == 0) { wr24(op, 1); return (size_t)(op + 3 - dst); }
    if (!zo_fctx_init(&f, cp)) { zo_fctx_free(&f); return ZO_ERROR; }
    f.rep[0] = 1; f.rep[1] = 4; f.rep[2] = 8;
    g_zo_win_start = 0;
    r = zo_frame_chunk(&f, cp, src, 0, n, 1, op);                                /* ZSTD_compressEnd: the whole input is one chunk */
    zo_fctx_free(&f);
    return r == ZO_ERROR ? ZO_ERROR : (size_t)(op - dst) + r;
}

/* ZSTD_compress2 with a CDict on a source above 128 KB, strategies ZSTD_fast and ZSTD_dfast: the COPY mode (the source is far above the
 * attach cut-offs) carried through ZSTD_compress_frameChunk — the context's tables start as copies of the CDict's and live across blocks,
 * every block runs the extDict parser with the dictionary as the other segment, the first block starts from the dictionary's repcodes
 * and entropy tables.  Restated for sources no longer than the window (the parameters are chosen for source + dictionary, so that is every
 * default case): then the dictionary stays valid for the whole frame (ZSTD_checkDictValidity, zstd_compress_internal.h:1140-1160). */
size_t zo_compress_frame_cdict(void* dstv, size_t cap, const void* srcv, size_t n, const zo_cdict* cd)
{
    uint8_t* const dst = (uint8_t*)dstv; const uint8_t* const src = (const uint8_t*)srcv;
    uint8_t* op = dst;
    zo_cparams cp, p; zo_fctx f; size_t r, i;
    if (n <= ZO_BLOCK_MAX) return zo_compress_unit_cdict(dstv, cap, srcv, n, cd);
    if (cd->cp.strategy > 2 || cap < zo_frame_bound(n)) return ZO_ERROR;
    {   int rc; g_zo_any_strategy = 1; rc = zo_get_cparams_mode(cd->level, n, cd->fullSize, 0, &p); g_zo_any_strategy = 0; if (rc < 0) return ZO_ERROR; }
    cp = cd->cp; cp.windowLog = p.windowLog;
    if (cd->len == 0) return zo_compress_frame_params(dstv, cap, srcv, n, &cp);
    /* zstd_compress.c:5153-5165: the CDict's TABLES are only used for a source below 128 KB or below six times the dictionary content
     * (a CDict made by ZSTD_createCDict; an "advanced" one always is, with default-level parameters requested).  Above that the reference
     * reloads the dictionary content into the context with the context's own parameters (ZSTD_compress_insertDictionary): not restated */
    if (n >= 6 * cd->fullSize) return ZO_ERROR;
    op += write_frame_header_dict(op, &cp, n, cd->dictID);
    if (!zo_fctx_init(&f, &cp)) { zo_fctx_free(&f); return ZO_ERROR; }
    for (i = 0; i < ((size_t)1 << cp.hashLog); i++) f.T[i] = cd->tabL[i] >> 8;                    /* zstd_compress.c:2379-2393 tags removed */
    if (cp.strategy == 2) for (i = 0; i < ((size_t)1 << cp.chainLog); i++) f.T[((size_t)1 << cp.hashLog) + i] = cd->tabS[i] >> 8;
    f.cd = cd; f.dictValid = 1; f.winLowIdx = 2;
    f.rep[0] = cd->rep[0]; f.rep[1] = cd->rep[1]; f.rep[2] = cd->rep[2];
    if (cd->hasEntropy) f.prev = cd->prev;
    r = zo_frame_chunk(&f, &cp, src, 0, n, 1, op);
    zo_fctx_free(&f);
    return r == ZO_ERROR ? ZO_ERROR : (size_t)(op - dst) + r;
}

/* ---- ZSTD_c_nbWorkers >= 1: one frame cut into jobs (zstdmt_compress.c).  Job k > 0 starts from a FRESH context that has loaded
 * the last `overlap` bytes of job k-1 as a raw-content prefix with ZSTD_dtlm_fast (every third position of the prefix goes into the
 * table(s), zstd_fast.c:50-86 / zstd_double_fast.c:56-90), with invalid repcodes {0,0,0} (:741) and no previous entropy tables; it
 * compresses its section in chunks of 512 KB, each one ZSTD_compressContinue call (:753-776; the block-split rule's `savings` is the
 * context's consumed - produced, header bytes included); the last job ends
 * the frame.  Inputs of at most 512 KB are compressed single-threaded (zstd_compress.c:6262). */
static void zo_fill_prefix(const zo_cparams* cp, const uint8_t* src, size_t p0, size_t p1, uint32_t* T)
{
    size_t ip;
    if (p1 - p0 <= 8) return;                                                    /* zstd_compress.c:4902 */
    {   size_t const maxDict = (size_t)8 << ((cp->hashLog > cp->chainLog ? cp->hashLog : cp->chainLog) < 28 ? (cp->hashLog > cp->chainLog ? cp->hashLog : cp->chainLog) : 28);
        if (p1 - p0 > maxDict) p0 = p1 - maxDict; }                              /* :4889-4896 */
    for (ip = p0; ip + 3 < (p1 - 8) + 2; ip += 3) {
        if (cp->strategy == 2) {
            T[((size_t)1 << cp->hashLog) + zo_hash(src + ip, cp->chainLog, cp->minMatch)] = (uint32_t)ip + 1;     /* small table */
            T[zo_hash(src + ip, cp->hashLog, 8)] = (uint32_t)ip + 1;                                                /* long table */
        } else T[zo_hash(src + ip, cp->hashLog, cp->minMatch)] = (uint32_t)ip + 1;
    }
}

/* the lazy strategies' part of ZSTD_loadDictionaryContent (zstd_compress.c:4878-4965) for a raw-content prefix [p0, p1): the window
 * starts at p0; of a prefix longer than 8 << max(hashLog, chainLog) only the suffix is indexed; EVERY position up to p1 - 8 goes in
 * (ZSTD_insertAndFindFirstIndex / ZSTD_row_update without its skip rule), then nextToUpdate = p1: the last 8 never do */
static void zo_lz_load_prefix(zo_lz* S, const zo_cparams* cp, const uint8_t* src, size_t p0, size_t p1)
{
    size_t ip = p0, idx;
    S->low = p0; S->hc.nextToUpdate = S->rw.nextToUpdate = p1;
    {   unsigned const big = cp->hashLog > cp->chainLog ? cp->hashLog : cp->chainLog;
        size_t const maxDict = (size_t)8 << (big < 28 ? big : 28);
        if (p1 - p0 > maxDict) ip = p1 - maxDict; }
    if (p1 - ip <= 8) return;
    if (S->useRow) zo_row_insert_range(&S->rw, src, ip, p1 - 8);
    else for (idx = ip; idx < p1 - 8; idx++) {
        uint32_t const h = zo_hash(src + idx, S->hc.hlog, S->hc.mls);
        S->hc.chain[idx & ((1u << S->hc.clog) - 1)] = S->hc.head[h];
        S->hc.head[h] = (uint32_t)idx + 1;
    }
}

size_t zo_mt_job_size(const zo_cparams* cp, unsigned long long jobSize)
{
    unsigned long long sec = jobSize;
    if (sec != 0 && sec < (512u << 10)) sec = 512u << 10;                        /* ZSTDMT_JOBSIZE_MIN */
    if (sec > (1024ull << 20)) sec = 1024ull << 20;                              /* ZSTDMT_JOBSIZE_MAX (64-bit) */
    if (sec == 0) { unsigned jl = cp->windowLog + 2; if (jl < 20) jl = 20; if (jl > 30) jl = 30; sec = 1ull << jl; }   /* :1168-1180 */
    return (size_t)sec;
}
size_t zo_mt_overlap_size(const zo_cparams* cp, int overlapLog)
{
    int const dflt = cp->strategy == 9 ? 9 : (cp->strategy >= 7 ? 8 : (cp->strategy >= 5 ? 7 : 6));   /* ZSTDMT_overlapLog_default (:1182-1203) */
    int const ov = overlapLog ? overlapLog : dflt;
    int const rlog = 9 - ov;
    int const ovLog = rlog >= 8 ? 0 : (int)cp->windowLog - rlog;
    return ovLog <= 0 ? 0 : (size_t)1 << ovLog;
}

size_t zo_compress_frame_mt_params(void* dstv, size_t cap, const void* srcv, size_t n, const zo_cparams* cp,
                                   unsigned long long jobSize, int overlapLog, int checksumFlag)
{
    uint8_t* const dst = (uint8_t*)dstv; const uint8_t* const src = (const uint8_t*)srcv;
    uint8_t* op = dst;
    size_t pos = 0, prevLen = 0, section, overlap;
    unsigned k = 0;
    if (cp->strategy < 1 || cp->strategy > 5 || cap < zo_frame_bound(n) + 4) return ZO_ERROR;
    if (cp->windowLog < 17 && n > ((size_t)1 << cp->windowLog)) return ZO_ERROR;
    if (n <= (512u << 10)) {                                                     /* single-threaded below ZSTDMT_JOBSIZE_MIN */
        size_t r = zo_compress_frame_params(dst, cap, src, n, cp);
        if (r == ZO_ERROR) return r;
        if (checksumFlag) { dst[4] |= 4; wr32(dst + r, (uint32_t)zo_xxh64(src, n, 0)); r += 4; }
        return r;
    }
    section = zo_mt_job_size(cp, jobSize); overlap = zo_mt_overlap_size(cp, overlapLog);
    if (section < overlap) section = overlap;                                    /* :1300 */
    op += write_frame_header(op, cp, n);
    if (checksumFlag) dst[4] |= 4;
    while (pos < n) {
        size_t const jLen = n - pos < section ? n - pos : section;
        int const lastJob = pos + jLen == n;
        size_t const preLen = k == 0 ? 0 : (prevLen < overlap ? prevLen : overlap);   /* :1404-1407 */
        zo_fctx f; size_t c0;
        if (!zo_fctx_init(&f, cp)) { zo_fctx_free(&f); return ZO_ERROR; }
        if (k == 0) { f.rep[0] = 1; f.rep[1] = 4; f.rep[2] = 8; g_zo_win_start = 0; }
        else {
            uint8_t hdr[18];
            f.rep[0] = f.rep[1] = f.rep[2] = 0; g_zo_win_start = pos - preLen;
            if (cp->strategy >= 3) zo_lz_load_prefix(&f.lz, cp, src, pos - preLen, pos);
            else zo_fill_prefix(cp, src, pos - preLen, pos, f.T);
            f.savings = -(long long)write_frame_header(hdr, cp, jLen);           /* the job's own (discarded) frame header counts as produced (:737-741, :4767) */
        }
        for (c0 = 0; c0 < jLen; c0 += (512u << 10)) {
            size_t const cLen = jLen - c0 < (512u << 10) ? jLen - c0 : (512u << 10);
            size_t const r = zo_frame_chunk(&f, cp, src, pos + c0, cLen, lastJob && c0 + cLen == jLen, op);
            if (r == ZO_ERROR) { zo_fctx_free(&f); g_zo_win_start = 0; return ZO_ERROR; }
            op += r;
            if (k == 0 && c0 == 0) f.savings -= (long long)(op - r - dst);       /* the first job wrote the real header in its first call */
        }
        zo_fctx_free(&f);
        prevLen = jLen; pos += jLen; k++;
    }
    g_zo_win_start = 0;
    if (checksumFlag) { wr32(op, (uint32_t)zo_xxh64(src, n, 0)); op += 4; }
    return (size_t)(op - dst);
}

size_t zo_compress_frame(void* dst, size_t cap, const void* src, size_t n, int level)
{
    zo_cparams cp;
    if (zo_get_cparams(level, n, &cp) != 0) return ZO_ERROR;
    return zo_compress_frame_params(dst, cap, src, n, &cp);
}

size_t zo_compress_unit(void* dst, size_t cap, const void* src, size_t n, int level)
{
    zo_cparams cp;
    if (zo_get_cparams(level, n, &cp) < 0) return ZO_ERROR;
    return zo_compress_unit_params(dst, cap, src, n, &cp);
}

size_t zo_compress_chunks(int level, size_t chunk, const void* src, size_t n, void* dst, size_t cap, size_t* sizes, size_t maxChunks)
{
    size_t off = 0, pos = 0, k = 0;
    if (n == 0) { size_t const r = zo_compress_unit(dst, cap, src, 0, level); if (sizes && maxChunks) sizes[0] = r; return r; }
    while (off < n) {
        size_t const len = n - off < chunk ? n - off : chunk;
        size_t const r = zo_compress_unit((uint8_t*)dst + pos, cap - pos, (const uint8_t*)src + off, len, level);
        if (r == ZO_ERROR) return ZO_ERROR;
        if (sizes && k < maxChunks) sizes[k] = r;
        k++; pos += r; off += len;
    }
    return pos;
}

/* ------------------------------------------------------------------ datagen (programs/datagen.c:45-153) */
static uint32_t rdg_rand(uint32_t* s)
{
    uint32_t r = *s; r *= 2654435761U; r ^= 2246822519U; r = (r << 13) | (r >> 19); *s = r; return r >> 5;
}
static uint32_t rdg_len(uint32_t* s) { if (rdg_rand(s) & 7) return rdg_rand(s) & 0xF; return (rdg_rand(s) & 0x1FF) + 0xF; }

void zo_datagen(void* buffer, size_t size, double matchProba, double litProba, unsigned seed)
{
    uint8_t ldt[8192]; uint8_t* const b = (uint8_t*)buffer; uint32_t s = seed; size_t pos = 0;
    uint32_t const mp32 = (uint32_t)(32768 * matchProba); uint32_t prevOffset = 1;
    memset(ldt, '0', sizeof(ldt));
    if (litProba <= 0.0) litProba = matchProba / 4.5;
    {   uint32_t ld = (uint32_t)(litProba * 256 + 0.001), u;
        uint8_t const first = ld ? '(' : 0, last = ld ? '}' : 255; uint8_t ch = ld ? '0' : 0;
        for (u = 0; u < 8192; ) {
            uint32_t const w = (((8192 - u) * ld) >> 8) + 1;
            uint32_t const end = u + w < 8192 ? u + w : 8192;
            while (u < end) ldt[u++] = ch;
            ch++; if (ch > last) ch = first;
        }
    }
    if (size == 0) return;
    while (matchProba >= 1.0) {
        size_t size0 = rdg_rand(&s) & 3;
        size0 = (size_t)1 << (16 + size0 * 2);
        size0 += rdg_rand(&s) & (size0 - 1);
        if (size < pos + size0) { memset(b + pos, 0, size - pos); return; }
        memset(b + pos, 0, size0); pos += size0;
        b[pos-1] = ldt[rdg_rand(&s) & 8191];
    }
    b[0] = ldt[rdg_rand(&s) & 8191]; pos = 1;
    while (pos < size) {
        if ((rdg_rand(&s) & 0x7FFF) < mp32) {
            uint32_t const length = rdg_len(&s) + 4;
            uint32_t const d = (uint32_t)(pos + length < size ? pos + length : size);
            uint32_t const repeatOffset = (rdg_rand(&s) & 15) == 2;
            uint32_t const randOffset = (rdg_rand(&s) & 0x7FFF) + 1;
            uint32_t const offset = repeatOffset ? prevOffset : (uint32_t)(randOffset < pos ? randOffset : pos);
            size_t match = pos - offset;
            while (pos < d) b[pos++] = b[match++];
            prevOffset = offset;
        } else {
            uint32_t const length = rdg_len(&s);
            uint32_t const d = (uint32_t)(pos + length < size ? pos + length : size);
            while (pos < d) b[pos++] = ldt[rdg_rand(&s) & 8191];
        }
    }
}
