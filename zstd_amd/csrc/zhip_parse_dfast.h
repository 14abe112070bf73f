// zhip_parse_dfast.h — gfx950 match finder for strategy ZSTD_dfast (levels 3-4), one wavefront per unit.
//
// WHAT it computes: exactly the sequences the reference's ZSTD_compressBlock_doubleFast_noDict_generic
// (lib/compress/zstd_double_fast.c:105-323) emits for a unit with no history (fresh tables, rep = {1,4,8}).
//
// HOW.  Same batch scheme as zhip_parse.h: the positions the reference would visit from the current point
// (ip, ip+step, ip+2*step, ... — zstd_double_fast.c:171-246; the gap grows every 256 bytes) are searched by the lanes
// of one wavefront at once, and the first event in the reference's own order (per position: repcode at ip+1, long
// match, short match) is found with ballots.  Differences that shape the kernel:
//   * two tables (8-byte "long" hash with hashLog bits, mls-byte "short" hash with chainLog bits).  For the 128 KB
//     parameter row they hold 2^16 + 2^15 entries — more than the 160 KB of LDS — so they live in HBM/L2 as plain
//     32-bit arrays, one private pair per unit; with no LDS table the kernel runs at full wave occupancy instead;
//   * lanes of one batch that hash alike (in either table) must see each other's inserts in lane order: two small LDS
//     scratch arrays (lane id written / read back at hash & 1023) flag the candidates, ballots make the exact groups;
//   * entries carry a 15-bit tag of the bytes the reference would compare, so only candidates that can match are fetched;
//   * a short match also looks at the long candidate of the NEXT position (:251-264) — lane K of a K-lane batch is a
//     helper that carries that position's bytes, long hash and candidate.
// All control flow is wave-uniform; the only LDS use is the 2 KB scratch.
#pragma once
#include <hip/hip_runtime.h>
#include "zhip_common.h"
#include "zhip_parse.h"

namespace zhip {

#ifndef ZHIP_DF_SCRATCH
#define ZHIP_DF_SCRATCH 1024u                       /* entries per scratch array */
#endif
__host__ __device__ inline uint32_t dfast_lds_bytes() { return 2u * ZHIP_DF_SCRATCH; }
// bytes of table memory one unit needs (long + short, 32-bit entries)
__host__ __device__ inline size_t dfast_table_bytes(uint32_t hashLog, uint32_t chainLog) { return ((size_t)4 << hashLog) + ((size_t)4 << chainLog); }

// Table entries are private to this kernel: position (17 bits) | 15-bit tag << 17.  The tag is a function of exactly the
// bytes the reference compares at a candidate (8 for the long table, 4 for the short one), so a tag mismatch proves the
// compare fails and the candidate's source bytes — a random HBM sector — need not be fetched at all.
#define ZHIP_DF_POS 0x1FFFFu
__device__ __forceinline__ uint32_t df_tag_long(uint32_t v /* mulhi64_top32(bytes, prime8) */) { return v & 0x7FFFu; }   // the hash's spare low bits
__device__ __forceinline__ uint32_t df_tag_short(uint32_t first4) { return (first4 * 2654435761U) >> 17; }

#define DF_POS(e)        (WIDE ? (e) : ((e) & ZHIP_DF_POS))
#define DF_TAGOK(e, tg)  (WIDE ? true : (((e) >> 17) == (tg)))
#define DF_ENTRY(p, tg)  (WIDE ? (p) : ((p) | ((tg) << 17)))

// exact groups of live lanes with equal key, given the lanes whose scratch slot was taken by another lane
__device__ __forceinline__ unsigned long long lane_groups(uint32_t key, unsigned long long losers, unsigned long long liveMask)
{
    unsigned long long grp = 0;
    while (losers) {
        int const j = first_lane(losers);
        uint32_t const kj = __builtin_amdgcn_readlane(key, j);
        unsigned long long const G = __ballot(key == kj) & liveMask;
        if (key == kj) grp = G;
        losers &= ~G;
    }
    return grp;
}


// ------------------------------------------------------------------ the dense-scan WINDOW of ZSTD_dfast (round 3)
// While the gap is 1 (the reference restarts at 1 after every match and keeps it for 256 bytes, zstd_double_fast.c:168-170, :232-236)
// every position is searched, so lane l takes position B+l: ONE gather from each table and ONE candidate fetch per 64 positions, and
// then EVERY event among them is resolved in the reference's order (per position: repcode at p+1, long match at p, short match at p
// with the long candidate of p+1) by mask arithmetic on E masks — per offset, which lanes equal the byte / the 4 bytes that offset back,
// one coalesced load per new offset — exactly like the ZSTD_fast window of zhip_parse.h.  The batch scheme below pays two HBM gathers
// and a candidate fetch per EVENT; dense-match data has five events per 64 positions.
// Lanes that hash alike (in either table) must see each other's inserts: the window ENDS in front of the first lane that shares a
// hash with an earlier one (W), so inside it every table entry is the one the reference would read and the inserts — collected in two
// masks, written once at the end — cannot collide.  A window that would end within its first lanes is not worth its gathers: the
// scan then uses the batch scheme (ZW_BATCH).  A match whose post-match inserts or immediate-repcode test fall beyond W hands them to
// the caller's round of loads (ZW_POST).
enum { ZW_POST = 3, ZW_BATCH = 4, ZW_CARRY = 5 };
#define ZHIP_DFW_NEED 80u            /* a window at B needs B + 80 <= n */
#define ZHIP_DFW_LANES 56u           /* events are taken from lanes below this */
#define ZHIP_DFW_MIN 12u             /* a window cut shorter than this by a hash collision is left to the batch scheme */
#ifndef ZHIP_DF_WINDOWS
#define ZHIP_DF_WINDOWS 1
#endif
#ifndef ZHIP_DF_DYNCUT
#define ZHIP_DF_DYNCUT 1             /* 0: round 5's window, cut once in front of the first lane that shares a hash with an earlier one */
#endif
#ifndef ZHIP_DFW_GATHER
#define ZHIP_DFW_GATHER 64u          /* lanes of a window that look their table entries up (the others count as beyond the cut W) */
#endif

// the second lowest lane among the lanes of `key` groups with two or more members (64: no group has two)
__device__ __forceinline__ uint32_t first_repeat_lane(uint32_t key, unsigned long long losers)
{
    uint32_t w = 64;
    while (losers) {
        int const j = first_lane(losers);
        uint32_t const kj = __builtin_amdgcn_readlane(key, j);
        unsigned long long const G = __ballot(key == kj);
        unsigned long long const rest = G & (G - 1);
        if (rest) { uint32_t const second = ff1u(rest); if (second < w) w = second; }
        losers &= ~G;
    }
    return w;
}

// Round 5.  The phase counters of round 4's window (profiles/r05_dfast_phases_before.log, Silesia-shaped mix, 35 000 cycles per window) put 17 %
// into the caller's round of loads behind a match that leaves the window (43 % of the windows ended that way), 13 % into waiting for the
// window's own source bytes behind the previous window's 128 scattered table stores (loads and stores retire in order), 10 % into the
// candidate fetch.  Two changes came of it, both exact and kept because they take loads off the chain:
//   * CARRY: a match that leaves the collision-free lanes hands its end to the NEXT window, which starts two positions in front of it — the
//     complementary inserts long[ip-2], short[ip-1] (zstd_double_fast.c:303-309) are lanes 0 and 1 of an ordinary window, the immediate-repcode
//     test (:313) is bit 2 of its E2 mask; the insert of curr+2 is a lane of the window that found the match, stored after the others.  An
//     immediate-repcode match that leaves the lanes carries the same way without the two inserts (carry 2).  No round of loads;
//   * the next window's source bytes are loaded BEFORE this window's table stores are issued (DfPre), unconditionally (a load inside a branch
//     is waited for inside the branch).
// Measured (profiles/r05_ab_dfast_carry_preload.log, 2 GiB per shape, the four on/off combinations): the stage's time does not move
// (Silesia-shaped 136.4-137.9 ms, text 166.0-169.6, datagen 121.0-121.2) — shortening a wavefront's chain buys nothing because the stage
// is bound by the RATE OF RANDOM MEMORY REQUESTS, not by their latency: profiles/r05_L3_silesia10_sq_tcc.txt has 5.1 G read + 1.0 G
// write requests at the L2's memory side per 136 ms = 45 G/s, which is what this machine's DRAM activates sustain; a wavefront that waits
// less only queues sooner.  Taking a tag match as a hit and letting the match's own E load confirm it (no candidate fetch: fewer
// requests) lost as well — the candidate fetch is also the prefetch that makes the E loads of the event loop hit (event loop 29.5 -> 48 M
// cycles per unit, profiles/r05_dfast_phases_tagtrust.log).  What would move the stage is fewer table requests per searched position:
// a window gathers 2 x 64 entries and the reference searches about a quarter of those positions on dense-match data.
// What one window hands to the next (B = ~0 / tB = ~0: nothing): the next window's source bytes (requested before the stores), and — round 5 —
// the table entries of the lanes this window gathered but did not reach.  Consecutive windows overlap (on text a window advances 29 of its 64
// positions: hash collisions cut it short) and the stage is bound by the NUMBER of random requests (profiles/README_r05.md): every gather
// misses the L2, the re-gather of the overlapping lanes included.  An entry read by the window at tB is still the table's entry when the next
// window starts iff no lane the window INSERTED shares its hash; the inserted lanes leave a mark in the scratch slots (the detector the window's
// front uses), a lane whose two slots carry no mark hands its entries and hit flags over (bit 2 of `hit`; the compares do not change: same
// position, same entry), the others — and slot aliases of inserted lanes — are gathered again.
struct DfPre { uint64_t bytes; uint32_t v1, v2, B; uint32_t eL, eS, hit, tB; };
#ifndef ZHIP_DF_PRELOAD
#define ZHIP_DF_PRELOAD 1            /* measurement switch: 0 = every window loads its own source bytes */
#endif
#ifndef ZHIP_DF_TABCARRY
#define ZHIP_DF_TABCARRY 1           /* measurement switch: 0 = every window gathers all of its lanes */
#endif

template <uint32_t MLS, bool WIDE>
__device__ __forceinline__ int window_dfast(const uint8_t* __restrict__ src, uint32_t n, uint32_t nm8, uint32_t shL, uint32_t shS,
                                            uint32_t* __restrict__ tabL, uint32_t* __restrict__ tabS, lds_u8* scrL, lds_u8* scrS, FastOut& out,
                                            uint32_t& ip_, uint32_t& anchor_, uint32_t& off1_, uint32_t& off2_, uint32_t& nextStep_,
                                            uint32_t prefixLow, uint32_t& curr_, bool& postFirst_, uint32_t& carry_, DfPre& pre)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t const carryIn = carry_;                                     // 1: lanes 0, 1 are ip-2, ip-1 of the previous window's last match; 2: the immediate-repcode test at lane 0 is due
    uint32_t const B = ip_ - (carryIn == 1 ? 2u : 0u), P = B + lane;
    if (out.pendLen) lits_flush(out);
    uint64_t bytes; uint32_t v1, v2;
    if (ZHIP_DF_PRELOAD && pre.B == B) { bytes = pre.bytes; v1 = pre.v1; v2 = pre.v2; }
    else {
        bytes = ld64(src + P);
        v1 = ld32(src + (P - off1_)); v2 = ld32(src + (P - off2_));       // an invalid repcode (0) reads the lane's own bytes
    }
    pre.B = ~0u;
    ZWPROF_SYNC(out, 1);
    uint32_t const cur32 = (uint32_t)bytes;
    uint32_t const vL = mulhi64_top32(bytes, 0xCF1BBCDCB7A56463ULL);
    uint32_t const hl = vL >> shL, hs = hash_pos<MLS>(bytes, shS);
    uint32_t const tgL = df_tag_long(vL), tgS = df_tag_short(cur32);
    // the lanes the window before already gathered (see DfPre) take their entries from its registers; they all read entry 0 — one line, one request
    bool haveT = false; uint32_t eLc = 0, eSc = 0, hitc = 0;
    if (ZHIP_DF_TABCARRY) {
        uint32_t const shift = B - pre.tB, ol = lane + shift;              // the lane's index in the window at tB
        eLc = pull(pre.eL, ol & 63u); eSc = pull(pre.eS, ol & 63u); hitc = pull(pre.hit, ol & 63u);
        haveT = pre.tB != ~0u && shift < 64u && ol < 64u && (hitc & 4u) != 0;
    }
    bool const gath = ZHIP_DFW_GATHER >= 64u || lane < ZHIP_DFW_GATHER;     // a lane beyond the gather width reads entry 0 (one line for all of them) and nothing is taken from it
#ifdef ZHIP_DBG_PRINT
    { unsigned long long hv = __ballot(haveT); if (lane == 0) printf("  dftab B=%u carried=%d\n", B, (int)__builtin_popcountll(hv)); }
#endif
    pre.tB = ~0u;
    uint32_t const tL = tabL[(haveT || !gath) ? 0u : hl], tS = tabS[(haveT || !gath) ? 0u : hs];
    uint32_t const eL = haveT ? eLc : tL, eS = haveT ? eSc : tS;
    ZWPROF_SYNC(out, 2);
    uint32_t const oldL = DF_POS(eL), oldS = DF_POS(eS);
    uint32_t const sl = hl & (ZHIP_DF_SCRATCH - 1), ss = hs & (ZHIP_DF_SCRATCH - 1);
    scrL[sl] = (uint8_t)lane; scrS[ss] = (uint8_t)lane;
    __builtin_amdgcn_wave_barrier();
    unsigned long long const loseL = __ballot(scrL[sl] != (uint8_t)lane), loseS = __ballot(scrS[ss] != (uint8_t)lane);
    __builtin_amdgcn_wave_barrier();
    // candidate bytes only where the entry's tag says they can match.  The fetch also brings the candidates' lines close: the E loads of
    // the event loop hit them (taking the tag's word and confirming by the E load alone was measured: the loads then miss one after the
    // other, event loop 29.5 -> 48 M cycles per unit, profiles/r05_dfast_phases_tagtrust.log)
    uint64_t cbL = ~bytes; uint32_t cbS = ~cur32;
    if (!haveT && gath && oldL != 0 && DF_TAGOK(eL, tgL)) cbL = ld64(src + (oldL < nm8 ? oldL : nm8));
    if (!haveT && gath && oldS != 0 && DF_TAGOK(eS, tgS)) cbS = ld32(src + (oldS < nm8 ? oldS : nm8));
    bool const hitL = haveT ? (hitc & 1u) != 0 : (oldL != 0 && oldL >= prefixLow && cbL == bytes);        // :203 (index >= lowest, ZSTD_selectAddr)
    bool const hitS = haveT ? (hitc & 2u) != 0 : (oldS != 0 && oldS >= prefixLow && cbS == cur32);        // :218
    ZWPROF_SYNC(out, 3);
    ZWPROF_COUNT(out, 10, 1);
#if ZHIP_DF_DYNCUT
    // THE CUT FROM THE SCAN POSITION ON (round 6; DESIGN 9.2 of round 5).  A lane's table entries are the ones the reference reads unless an EARLIER lane of the same hash
    // (in either table) is inserted before the scan gets to it — it is, when INSL / INSS hold it already or when it lies at or behind the scan position i (every lane from i
    // on is inserted as the scan passes it, until the next event: and the next event is what is being looked for).  A member INSIDE a match is neither: the scan jumps over it.
    // Round 5 cut the window once, in front of the first lane that shares a hash with ANY earlier lane, and left windows cut before their 12th lane to the batch scheme: 575 of
    // a unit's 3 150 scan steps on the Silesia-shaped mix, 22 % of the stage — most of them runs and long matches, where the colliding lanes lie inside the match lane 0 finds.
    // prevL / prevS: the earlier lanes of this lane's exact hash groups (the scratch slots only flag candidates: aliases of different hashes fall out in lane_groups).
    unsigned long long gL = 0, gS = 0;
    if (loseL) gL = lane_groups(hl, loseL, ~0ull);
    if (loseS) gS = lane_groups(hs, loseS, ~0ull);
    unsigned long long const prevL = gL & lanes_below(lane), prevS = gS & lanes_below(lane);
    bool const anyGroups = __ballot((prevL | prevS) != 0) != 0;
    uint32_t W = ZHIP_DFW_GATHER < 64u ? ZHIP_DFW_GATHER : 64u;                  // first lane from i on that the window cannot look up (recomputed when the scan gets there: it only moves up)
    uint32_t const Wmax = W;
#define DW_CUT(i_) (anyGroups ? umin32(Wmax, ff1u(__ballot(((prevL & (INSL | lanes_from(i_))) | (prevS & (INSS | lanes_from(i_)))) != 0) & lanes_from(i_))) : Wmax)
    carry_ = 0;
#else
    uint32_t W = ZHIP_DFW_GATHER < 64u ? ZHIP_DFW_GATHER : 64u;
    if (loseL) { uint32_t const w = first_repeat_lane(hl, loseL); if (w < W) W = w; }
    if (loseS) { uint32_t const w = first_repeat_lane(hs, loseS); if (w < W) W = w; }
#ifdef ZHIP_DBG_PRINT
    if (W < ZHIP_DFW_MIN && lane == 0) printf("  dfwin B=%u carryIn=%u W=%u -> ZW_BATCH\n", B, carryIn, W);
#endif
    if (W < ZHIP_DFW_MIN) { ZWPROF_COUNT(out, 14, 1); ZWPROF(out, 4); return ZW_BATCH; }     // (carry_ stays: the caller settles it by loads)
    carry_ = 0;
    // events from lanes below this: an event at lane j reads the LONG candidate of lane j+1 (:251) and inserts lane j+1 (:283), both must
    // lie below W; every other inserted lane is checked where it arises
    uint32_t const hiBound = W - 1 < ZHIP_DFW_LANES ? W - 1 : ZHIP_DFW_LANES;
#endif
    unsigned long long const ML = __ballot(hitL), MS = __ballot(hitS), L1 = __ballot(hitL && oldL > prefixLow);   // :260 long match at ip+1: index > lowest
    uint32_t const x1 = off1_ ? cur32 ^ v1 : 1u, x2 = off2_ ? cur32 ^ v2 : 1u;
    unsigned long long E1q = __ballot(x1 == 0), E1b = __ballot((x1 & 0xFFu) == 0);
    unsigned long long E2q = __ballot(x2 == 0), E2b = __ballot((x2 & 0xFFu) == 0);

    uint32_t anchor = anchor_, off1 = off1_, off2 = off2_, nextStep = nextStep_;
    uint32_t const nbSeq0 = out.nbSeq, anchorEntry = anchor_;
    uint32_t evA = 0, evB = 0, nEv = ~0u;
    unsigned long long INSL = 0, INSS = 0, COV = 0;
    uint32_t backBefore = 0, sumLit = 0, i = 0, lateLane = 64;
    bool fresh = false;
    int status = ZW_CONT;
    ZWPROF(out, 4);
#define DW_EMIT(ll, ob, ml) do { uint32_t const mb_ = (ml) - 3;                                                   \
        if (((ll) | mb_) > 0xFFFF) {                                                                              \
            if ((ll) > 0xFFFF) { out.longType = 1; out.longPos = out.nbSeq; }                                     \
            if (mb_ > 0xFFFF) { out.longType = 2; out.longPos = out.nbSeq; } }                                    \
        uint32_t const slot_ = out.nbSeq - nbSeq0;                                                                \
        evA = ZHIP_WRITELANE((ob), slot_, evA); evB = ZHIP_WRITELANE(((ll) & 0xFFFFu) | (mb_ << 16), slot_, evB); \
        out.nbSeq++; } while (0)
    // :313-327 the immediate-repcode loop from lane e_ on; leave_: it went on beyond the collision-free lanes
#define DW_IMMEDIATE(e_, leave_) do {                                                                             \
        while (off2 > 0 && (((e_) < 64 ? E2q >> (e_) : 0ull) & 1)) {                                              \
            uint32_t const rl = 4 + fwd_run(src, nm8, B, E2b, (e_) + 4, off2);                                    \
            {   uint32_t const t = off2; off2 = off1; off1 = t; }                                                 \
            {   unsigned long long t = E2q; E2q = E1q; E1q = t; t = E2b; E2b = E1b; E1b = t; }                    \
            INSL |= 1ull << (e_); INSS |= 1ull << (e_);                                                           \
            DW_EMIT(0u, 1u, rl);                                                                                  \
            uint32_t const en = (e_) + rl;                                                                        \
            COV |= en < 64 ? ZHIP_SBFM64(rl, (e_)) : lanes_from(e_);                                              \
            (e_) = en; anchor = B + (e_); nextStep = B + (e_) + 256; fresh = true;                               \
            if ((e_) >= (ZHIP_DF_DYNCUT ? 64u : W)) { (leave_) = true; break; }     /* the next test reads beyond the lanes at hand */ \
        } } while (0)
    // a scan that leaves the collision-free lanes behind a match: the next window takes it over when there is room for one, else the caller's loads
#define DW_LEAVE(e_, kind_, currLane_) do {                                                                       \
        i = (e_);                                                                                                 \
        if (B + (e_) - ((kind_) == 1 ? 2u : 0u) + ZHIP_DFW_NEED <= n) { carry_ = (kind_); status = ZW_CARRY; } \
        else { status = ZW_POST; postFirst_ = (kind_) == 1; curr_ = (kind_) == 1 ? B + (currLane_) : 0u; } } while (0)
    if (carryIn) {
        uint32_t e = 0;
        if (carryIn == 1) { INSL |= 1ull; INSS |= 2ull; e = 2; }         // :305-309 long[ip-2], short[ip-1]
        bool leave = false;
        DW_IMMEDIATE(e, leave);
        i = e;
        if (leave) { DW_LEAVE(e, 2u, 0u); goto dw_done; }
    }
#if ZHIP_DF_DYNCUT
    W = DW_CUT(i);
#endif
    for (;;) {
        // lanes searched with gap 1: those whose position + 1 stays below nextStep (:232); the entry scan may reach it inside the window
        uint32_t const kLane = (int32_t)(nextStep - B) > 64 ? 64u : ((int32_t)(nextStep - B) < 0 ? 0u : nextStep - B);
#if ZHIP_DF_DYNCUT
        if (i + 1 >= W && W < Wmax) W = DW_CUT(i);                           // the scan has reached the cut: where is it from here?
        // lanes i .. hiS-1 take any event (an event at lane j reads the LONG candidate of lane j+1, :251, and inserts lane j+1, :283: j+1 < W); lane W-1 itself takes a
        // repcode or a long match (neither looks at lane W's entries; what they insert beyond W is put in order by the winners' write at the end) but not a short one
        uint32_t const limitLane = kLane < ZHIP_DFW_LANES ? kLane : ZHIP_DFW_LANES;
        uint32_t const hiS = limitLane < W - 1 ? limitLane : W - 1;
        bool const lastOK = W - 1 < limitLane && i <= W - 1;                // lane W-1 is within the gap and the lanes events are taken from
        if (i >= hiS && !lastOK) {
            if (i >= kLane && kLane <= ZHIP_DFW_LANES && !fresh) status = ZW_INC;    // (i == kLane: the position nextStep-1 was the last at gap 1)
            else status = ZW_CONT;                                           // the scan (its nextStep rides along) goes on in the next window
            break;
        }
        unsigned long long const span = i < hiS ? ZHIP_SBFM64(hiS - i, i) : 0ull;
        unsigned long long const R = (E1q >> 1) & span;                      // bit l: the repcode test of iteration l (position l+1, :190) hits
        unsigned long long any = R | ((ML | MS) & span);
        if (any == 0) {
            INSL |= span; INSS |= span;
            i = hiS;
            if (!lastOK) continue;                                           // (the exit test above decides how the scan goes on)
            // lane W-1 (= i now): its own entries are good; a short match there would read lane W's long candidate
            unsigned long long const bitW = 1ull << i;
            if ((((E1q >> 1) | ML) & bitW) != 0) any = bitW;
            else if (MS & bitW) { status = ZW_BATCH; ZWPROF_COUNT(out, 14, 1); break; }      // the batch scheme takes this one event (its lanes see each other's inserts)
            else { INSL |= bitW; INSS |= bitW; i++; continue; }             // no event: searched and inserted like any lane; the scan now stands in front of W
        }
#else
        uint32_t const hiS = kLane < hiBound ? kLane : hiBound;
        if (i >= hiS) {
            if (kLane <= hiBound && i >= kLane && !fresh) status = ZW_INC;    // (i == kLane: the position nextStep-1 was the last at gap 1)
            else status = ZW_CONT;                                           // the scan (its nextStep rides along) goes on in the next window
            break;
        }
        unsigned long long const span = ZHIP_SBFM64(hiS - i, i);
        unsigned long long const R = (E1q >> 1) & span;                      // bit l: the repcode test of iteration l (position l+1, :190) hits
        unsigned long long const any = R | ((ML | MS) & span);
        if (any == 0) {
            INSL |= span; INSS |= span;
            i = hiS;
            continue;                                                        // (the exit test above decides how the scan goes on)
        }
#endif
        uint32_t const j = ff1u(any);
        int const kind = (((ZHIP_DF_DYNCUT ? (E1q >> 1) : R) >> j) & 1) ? 1 : (((ML >> j) & 1) ? 2 : 3);
        {   unsigned long long const done = ZHIP_SBFM64(j + 1 - i, i);       // :187 both tables updated for every position up to the event's
            INSL |= done; INSS |= done; }
        uint32_t s, e, back = 0, offBase;
        if (kind == 1) {
            s = j + 1;
            uint32_t const fl = fwd_run(src, nm8, B, E1b, s + 4, off1);
            e = s + 4 + fl; offBase = 1;
        } else {
            uint32_t off, cand;
            unsigned long long Wq, Wb;                                        // the masks of the match's offset
            if (kind == 2) {
                cand = __builtin_amdgcn_readlane(oldL, (int)j); off = B + j - cand;
                uint32_t x = 1;
                ZWPROF(out, 5);
                if (P >= off) x = cur32 ^ ld32(src + (P - off));
                Wq = __ballot(x == 0); Wb = __ballot((x & 0xFFu) == 0);
                ZWPROF_SYNC(out, 15);
                s = j;
                e = j + 8 + fwd_run(src, nm8, B, Wb, j + 8, off);
            } else {
                uint32_t const candS = __builtin_amdgcn_readlane(oldS, (int)j), offS = B + j - candS;
                bool const long1 = (L1 >> (j + 1)) & 1;
                uint32_t const cand1 = __builtin_amdgcn_readlane(oldL, (int)(j + 1)), offL = B + j + 1 - cand1;
                uint32_t xs = 1, xl = 1;
                ZWPROF(out, 5);
                if (P >= offS) xs = cur32 ^ ld32(src + (P - offS));
                if (long1 && P >= offL) xl = cur32 ^ ld32(src + (P - offL));
                unsigned long long const Sq = __ballot(xs == 0), Sb = __ballot((xs & 0xFFu) == 0);
                unsigned long long const Lq = __ballot(xl == 0), Lb = __ballot((xl & 0xFFu) == 0);
                ZWPROF_SYNC(out, 15);
                uint32_t const mS = 4 + fwd_run(src, nm8, B, Sb, j + 4, offS);
                uint32_t const mL = long1 ? 8 + fwd_run(src, nm8, B, Lb, j + 9, offL) : 0;
                if (mL > mS) { s = j + 1; e = s + mL; off = offL; cand = cand1; Wq = Lq; Wb = Lb; }     // :251-264 the long match at ip+1 wins
                else { s = j; e = s + mS; off = offS; cand = candS; Wq = Sq; Wb = Sb; }
            }
            INSL |= 1ull << (j + 1);                                          // :283-291 hashLong[hl1] = ip1 (gap 1 < 4)
            // catch up (:207, :267): equal bytes in front of the match, as far as the literals and the window's low end allow
            uint32_t const room = B + s - anchor;
            uint32_t const limit = room < cand - prefixLow ? room : cand - prefixLow;
            uint32_t run = 0;
            if (s) { unsigned long long const t = ~Wb << (64 - s); run = t ? (uint32_t)__clzll((long long)t) : s; }
            if (run == s && limit > run) run += wave_count_back(src, B, B - off, limit - run);
            back = run < limit ? run : limit;
            off2 = off1; E2q = E1q; E2b = E1b; off1 = off; E1q = Wq; E1b = Wb;
            offBase = off + 3;
        }
        {   uint32_t const ll = B + s - anchor - back;
            DW_EMIT(ll, offBase, e - s + back);
            sumLit += ll; }
        uint32_t sL = s - back;
        if (back > s) { backBefore = back - s; sL = 0; }
        anchor = B + e;
        fresh = true; nextStep = B + e + 256;
        if (ZHIP_DF_DYNCUT ? e >= 64 : (e >= W || j + 2 >= W)) {              // the scan leaves the window (round 5: the collision-free lanes) behind this match
            COV |= e < 64 ? ZHIP_SBFM64(e - sL, sL) : lanes_from(sL);
            DW_LEAVE(e, 1u, j);
            if (status == ZW_CARRY) {                                         // :303-307 curr+2 is a lane of this window (j <= 55); beyond W it is stored after the others
                if (ZHIP_DF_DYNCUT || j + 2 < W) { INSL |= 1ull << (j + 2); INSS |= 1ull << (j + 2); }
                else lateLane = j + 2;
            }
            break;
        }
        COV |= ZHIP_SBFM64(e - sL, sL);
        INSL |= (1ull << (j + 2)) | (1ull << (e - 2));                        // :305-310 complementary insertion
        INSS |= (1ull << (j + 2)) | (1ull << (e - 1));
        bool leave = false;
        DW_IMMEDIATE(e, leave);
        i = e;
        if (leave) { DW_LEAVE(e, 2u, 0u); break; }
    }
dw_done:
#undef DW_EMIT
#undef DW_IMMEDIATE
#undef DW_LEAVE
#ifdef ZHIP_DBG_PRINT
    if (lane == 0) printf("  dfwin B=%u carryIn=%u W=%u hiBound=%u -> i=%u status=%d carry=%u INSL=%llx INSS=%llx ML=%llx MS=%llx COV=%llx nbSeq=%u off=%u/%u nextStep=%u late=%u\n", B, carryIn, W, (uint32_t)(ZHIP_DF_DYNCUT ? 0u : 1u), i, status, carry_, INSL, INSS, ML, MS, COV, out.nbSeq, off1, off2, nextStep, lateLane);
#endif
    ZWPROF_SYNC(out, 5);
    ZWPROF_COUNT(out, 11, out.nbSeq - nbSeq0);
    if (status == ZW_POST) ZWPROF_COUNT(out, 13, 1);
    // the next window's source bytes, requested before this window's stores (loads and stores retire in order: behind the 128 scattered table
    // stores they would wait for every one of them)
    {   uint32_t const nB = B + i - (carry_ == 1 ? 2u : 0u);
        bool const nxt = (status == ZW_CONT || status == ZW_CARRY) && nB + ZHIP_DFW_NEED <= n;
        // unconditional on purpose (a load inside a branch is waited for inside the branch): without a next window the lanes read their own bytes again
        uint32_t const q = (nxt ? nB : B) + lane;
        if (ZHIP_DF_PRELOAD) {
            pre.bytes = ld64(src + q);
            pre.v1 = ld32(src + (q - (nxt ? off1 : 0u))); pre.v2 = ld32(src + (q - (nxt ? off2 : 0u)));
            pre.B = nxt ? nB : ~0u;
        }
        if (ZHIP_DF_TABCARRY && nxt && nB < B + 64u) {
            // which of the lanes the next window shares with this one still hold the table's entries: those no inserted lane shares a slot with
            unsigned long long const insAnyL = INSL | (lateLane < 64 ? 1ull << lateLane : 0ull), insAnyS = INSS | (lateLane < 64 ? 1ull << lateLane : 0ull);
            if (__builtin_amdgcn_inverse_ballot_w64(insAnyL)) scrL[sl] = 0x80;
            if (__builtin_amdgcn_inverse_ballot_w64(insAnyS)) scrS[ss] = 0x80;
            __builtin_amdgcn_wave_barrier();
            bool const clean = scrL[sl] != 0x80 && scrS[ss] != 0x80;
            __builtin_amdgcn_wave_barrier();
            pre.eL = eL; pre.eS = eS; pre.hit = (hitL ? 1u : 0u) | (hitS ? 2u : 0u) | ((clean && (haveT || gath)) ? 4u : 0u);
            pre.tB = B;
        }
    }
#if ZHIP_DF_DYNCUT
    // the window's table writes: of the inserted lanes of one hash the HIGHEST writes (the reference's inserts of a window follow each other in position order), all at once
    if (((INSL >> lane) & 1) && (gL & INSL & (lanes_from(lane) << 1)) == 0) tabL[hl] = DF_ENTRY(P, tgL);
    if (((INSS >> lane) & 1) && (gS & INSS & (lanes_from(lane) << 1)) == 0) tabS[hs] = DF_ENTRY(P, tgS);
#undef DW_CUT
#else
    // the window's table writes (no two inserted lanes share a hash: they all lie below W)
    if (__builtin_amdgcn_inverse_ballot_w64(INSL)) tabL[hl] = DF_ENTRY(P, tgL);
    if (__builtin_amdgcn_inverse_ballot_w64(INSS)) tabS[hs] = DF_ENTRY(P, tgS);
#endif
    __builtin_amdgcn_wave_barrier();
    if (lateLane < 64) {                                                     // curr+2 of a carried match: the highest position this window inserts, so it is stored last
        if (lane == lateLane) { tabL[hl] = DF_ENTRY(P, tgL); tabS[hs] = DF_ENTRY(P, tgS); }
        __builtin_amdgcn_wave_barrier();
    }
    nEv = out.nbSeq - nbSeq0;
    if (lane < nEv) {
        ZhipSeq q; q.offBase = evA; q.litLength = (uint16_t)evB; q.mlBase = (uint16_t)(evB >> 16);
        out.seqs[nbSeq0 + lane] = q;
    }
    {   unsigned long long LIT = i < 64 ? (~COV & lanes_below(i)) : ~COV;
        if (carryIn == 1) LIT &= ~3ull;                                      // lanes 0, 1 lie in front of the anchor
        if (__builtin_amdgcn_inverse_ballot_w64(LIT)) {
            uint32_t const before = __builtin_amdgcn_mbcnt_hi((uint32_t)(COV >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)COV, 0));
            out.lits[out.litPos + (B - anchorEntry) + lane - before - backBefore] = (uint8_t)bytes;
        }
    }
    out.litPos += sumLit;
    ip_ = B + i; anchor_ = anchor; off1_ = off1; off2_ = off2; nextStep_ = nextStep;
    ZWPROF(out, 6);
    return status;
}

// ONE round of loads for everything a dfast match needs (round 3; before: up to three dependent wave_count_* calls): lanes 0..31 compare
// 8 bytes each FORWARD from the main candidate (256 B), lanes 32..47 forward from the long candidate of the next position (128 B,
// zstd_double_fast.c:251-264), lanes 48..55 BACKWARD from the main pair, lanes 56..63 backward from the other pair (64 B each; which
// pair wins is only known once both forward lengths are).  A run that fills its lanes goes on with the loops of zhip_parse.h.
struct DfExt { uint32_t fwdM, fwdL, backM, backL; };
__device__ __forceinline__ DfExt df_extend(const uint8_t* src, uint32_t nm8, uint32_t posM, uint32_t candM, bool useL, uint32_t posL, uint32_t candL,
                                            bool useBack, uint32_t bposM, uint32_t bcandM, uint32_t limM, uint32_t bposL, uint32_t bcandL, uint32_t limL)
{
    uint32_t const lane = (uint32_t)lane_id();
    bool const isF = lane < 48, isFL = lane >= 32 && lane < 48, isBL = lane >= 56;
    uint32_t same;
    if (isF) {
        uint32_t const k = isFL ? lane - 32 : lane;
        uint32_t const q = (isFL ? posL : posM) + 8u * k, off = isFL ? posL - candL : posM - candM;
        same = (isFL && !useL) ? 0u : lane_same_fwd(src, q, off, nm8);
    } else {
        uint32_t const j8 = 8u * (isBL ? lane - 56 : lane - 48);
        uint32_t const mp = isBL ? bposL : bposM, cd = isBL ? bcandL : bcandM, lim = isBL ? limL : limM;
        bool const on = useBack && (!isBL || useL) && lim > j8;
        uint32_t const rr = lim - j8, r = on ? (rr < 8 ? rr : 8) : 0;        // bytes of this lane's chunk: the r bytes that end at mp - j8
        uint32_t const qb = r ? mp - j8 - r : mp;
        uint64_t const x = r ? (ld64(src + qb) ^ ld64(src + (qb - (mp - cd)))) : 0;
        uint64_t const y = x << (8 * ((8 - r) & 7));                          // byte r-1 (closest to mp) -> top byte
        uint32_t const sm = y ? (uint32_t)__clzll((long long)y) >> 3 : r;
        same = r ? sm : 0;
    }
    unsigned long long const stop = __ballot(same < 8);
    DfExt e;
    {   unsigned long long const m = stop & 0xFFFFFFFFull;
        if (m) { int const f = first_lane(m); e.fwdM = 8u * (uint32_t)f + __builtin_amdgcn_readlane(same, f); }
        else e.fwdM = 256 + wave_count_fwd(src, posM + 256, candM + 256, nm8); }
    e.fwdL = 0;
    if (useL) {
        unsigned long long const m = (stop >> 32) & 0xFFFFull;
        if (m) { int const f = first_lane(m); e.fwdL = 8u * (uint32_t)f + __builtin_amdgcn_readlane(same, f + 32); }
        else e.fwdL = 128 + wave_count_fwd(src, posL + 128, candL + 128, nm8);
    }
    e.backM = e.backL = 0;
    if (useBack) {
        {   unsigned long long const m = (stop >> 48) & 0xFFull;
            if (m) { int const f = first_lane(m); e.backM = 8u * (uint32_t)f + __builtin_amdgcn_readlane(same, f + 48); }
            else e.backM = 64 + wave_count_back(src, bposM - 64, bcandM - 64, limM - 64); }
        if (useL) {
            unsigned long long const m = stop >> 56;
            if (m) { int const f = first_lane(m); e.backL = 8u * (uint32_t)f + __builtin_amdgcn_readlane(same, f + 56); }
            else e.backL = 64 + wave_count_back(src, bposL - 64, bcandL - 64, limL - 64);
        }
    }
    return e;
}

// One block of ZSTD_dfast over src[b0, n) with the two tables as the previous blocks of the same frame left them (a unit: b0 = 0,
// fresh tables).  WIDE: entries are plain 32-bit positions (a frame's positions exceed 17 bits) — no tag, every nonzero candidate
// is fetched; else `position | tag << 17`.  Candidates must lie at or above prefixLow; the reference is asymmetric about the bound
// itself: long / short match at ip: index >= lowest (ZSTD_selectAddr, zstd_double_fast.c:200, :214), the long match at ip+1:
// index > lowest (:260), backward extension: match > lowest (:207, :267).
template <uint32_t MLS, bool WIDE>
__device__ inline void parse_dfast_block(const uint8_t* __restrict__ src, uint32_t b0, uint32_t n, uint32_t prefixLow, uint32_t maxRep,
                                         uint32_t repIn1, uint32_t repIn2, uint32_t repIn3, const ZhipUnit& u, unsigned char* smem,
                                         uint32_t* __restrict__ tabL, uint32_t* __restrict__ tabS,
                                         ZhipSeq* seqs, uint8_t* lits, ZhipParse* meta)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t const shL = 32 - u.hashLog, shS = 32 - u.chainLog;
    FastOut out; out.seqs = seqs; out.lits = lits; out.nbSeq = 0; out.longPos = 0; out.longType = 0;
    out.litPos = 0; out.pendV = 0; out.pendSh = 0; out.pendOff = 0; out.pendLen = 0;
    ZPROF_DECL
#ifdef ZHIP_PROF
    out.zp = zp_acc_; out.zlast = &zp_last_;
#endif
    ZPROF(0);
    lds_u8* const scrL = (lds_u8*)(uintptr_t)smem;
    lds_u8* const scrS = (lds_u8*)(uintptr_t)(smem + ZHIP_DF_SCRATCH);

    uint32_t anchor = b0, off1 = repIn1, off2 = repIn2, saved1 = 0, saved2 = 0;
    // :158-164  a repcode that reaches below the window is set aside for the block
    if (off2 > maxRep) { saved2 = off2; off2 = 0; }
    if (off1 > maxRep) { saved1 = off1; off1 = 0; }

    if (n - b0 >= 9) {                      // shorter blocks never run an iteration (ip1 = ip + 1 <= n - 8); keeps n - 8 >= b0
    uint32_t const nm8 = n - 8;
    int32_t const ilimit = (int32_t)nm8;
    uint32_t ip = b0 + (b0 == prefixLow);                                    // :157
    if (ip != b0 && lane == 0) lits[0] = src[b0];                            // that position is never searched: windows only store their own lanes' literals
    // Batch width.  Every searched lane costs two random table gathers (HBM/L2 sectors), and everything after the first
    // event of a batch is thrown away, so the batch is only as wide as events have recently been far apart: the running
    // mean distance (x16 fixed point) + 4 (best of the sweep in scripts/df_sweep.sh), doubled after a batch without an event.  Any width is exact.
    uint32_t const kMul = u.pad0 ? (uint32_t)(u.pad0 >> 4) : 8u, kAdd = u.pad0 ? (uint32_t)(u.pad0 & 15) : 4u;   // width = mean * kMul/8 + kAdd (measurement knob, any value is exact)
    uint32_t evAvg16 = 12u << 4, kCap = 32;
    bool have = false; uint64_t nbytes = 0; uint32_t nrv = 0;                // the source bytes of the next batch, when the round behind a match fetched them
    uint32_t carry = 0;                                                      // a window handed the end of its last match to the next one (window_dfast)
    DfPre pre; pre.bytes = 0; pre.v1 = 0; pre.v2 = 0; pre.B = ~0u;            // what one window hands to the next: source bytes, table entries (DfPre)
    pre.eL = 0; pre.eS = 0; pre.hit = 0; pre.tB = ~0u;
    for (;;) {                                                               // one turn per match (:167)
        uint32_t step = 1, nextStep = ip + 256;
        if ((int32_t)(ip + 1) > ilimit) break;                               // :172
        int evKind = 0;                      // 0 none (unit finished), 1 repcode, 2 long, 3 short
        uint32_t curr = 0, candE = 0, ip1 = 0, cand1 = 0; bool long1 = false;
        int winDone = 0;                     // 2: a window emitted a match and the round of loads behind it is still due
        bool postFirst = true, skipCurr2 = false;
        // windows while the gap is 1 and the unit has room; a scan that meets a hash collision early goes on in batches
        while (ZHIP_DF_WINDOWS && ip - (carry == 1 ? 2u : 0u) + ZHIP_DFW_NEED <= n) {
            int const st = window_dfast<MLS, WIDE>(src, n, nm8, shL, shS, tabL, tabS, scrL, scrS, out, ip, anchor, off1, off2, nextStep, prefixLow, curr, postFirst, carry, pre);
            have = false;
            if (st == ZW_CONT || st == ZW_CARRY) continue;                   // (a carrying window has checked that the next one has room)
            pre.B = ~0u; pre.tB = ~0u;
            if (st == ZW_POST) { winDone = 2; break; }
            if (st == ZW_BATCH && carry) {                                   // a hash collision in the first lanes of a carried window: what was carried goes by loads
                winDone = 2; postFirst = carry == 1; skipCurr2 = true; curr = 0; carry = 0;
                break;
            }
            if (st == ZW_INC) { step = 2; nextStep += 256; }
            break;                                                           // ZW_INC / ZW_BATCH: the batch scheme takes over
        }
        if (winDone == 0) {
        for (;;) {
            // lanes 0..K-1 search p_j = ip + j*step; lane K is the helper for position p_K (= ip1 of lane K-1)
            uint32_t const p = ip + lane * step;
            bool const inc = (int32_t)(p + step) >= (int32_t)nextStep;                    // :232 step++ after this position
            bool const endAfter = (int32_t)(p + 2 * step + (inc ? 1u : 0u)) > ilimit;     // :246 next position does not run
            unsigned long long const mStop = __ballot(inc || endAfter);
            int K = mStop ? first_lane(mStop) + 1 : 64;
            if (K > 63) K = 63;
            if (K > (int)kCap) K = (int)kCap;
            bool const lastInc = (__ballot(inc) >> (K - 1)) & 1, lastEnd = (__ballot(endAfter) >> (K - 1)) & 1;
            unsigned long long const liveMask = below_mask(K + 1), searchMask = below_mask(K);
            bool const live = (int)lane <= K;

            uint32_t const pc = p < nm8 ? p : nm8;
            uint64_t bytes; uint32_t rv;
            if (have) { bytes = nbytes; rv = nrv; have = false; }            // loaded with the round behind the previous match (step is 1 there)
            else { bytes = ld64(src + pc); rv = ld32(src + (pc + 1 - off1)); }   // off1 <= pc always; off1 == 0 is masked below
            uint32_t const vL = mulhi64_top32(bytes, 0xCF1BBCDCB7A56463ULL);
            uint32_t const hl = vL >> shL, hs = hash_pos<MLS>(bytes, shS);
            uint32_t const tgL = df_tag_long(vL), tgS = df_tag_short((uint32_t)bytes);
            uint32_t const eL = live ? tabL[hl] : 0, eS = live ? tabS[hs] : 0;
            uint32_t const oldL = DF_POS(eL), oldS = DF_POS(eS);
            uint32_t const sl = hl & (ZHIP_DF_SCRATCH - 1), ss = hs & (ZHIP_DF_SCRATCH - 1);
            if (live) { scrL[sl] = (uint8_t)lane; scrS[ss] = (uint8_t)lane; }
            __builtin_amdgcn_wave_barrier();
            unsigned long long const loseL = __ballot(live && scrL[sl] != (uint8_t)lane);
            unsigned long long const loseS = __ballot(live && scrS[ss] != (uint8_t)lane);
            __builtin_amdgcn_wave_barrier();

            // candidate bytes are only fetched where the entry's tag says they can match (a random sector each)
            uint64_t cbL = ~bytes; uint32_t cbS = ~(uint32_t)bytes;
            if (oldL != 0 && DF_TAGOK(eL, tgL)) cbL = ld64(src + (oldL < nm8 ? oldL : nm8));     // table values are <= n-8 by construction
            if (oldS != 0 && DF_TAGOK(eS, tgS)) cbS = ld32(src + (oldS < nm8 ? oldS : nm8));
            uint32_t candL = oldL, candS = oldS;
            unsigned long long grpL = 0, grpS = 0;
            if (loseL) {
                grpL = lane_groups(hl, loseL, liveMask);
                unsigned long long const prev = grpL & below_mask((int)lane);
                uint32_t const pd = prev ? 63u - (uint32_t)__clzll((long long)prev) : lane;
                uint32_t const dp = __shfl(p, (int)pd), dlo = __shfl((uint32_t)bytes, (int)pd), dhi = __shfl((uint32_t)(bytes >> 32), (int)pd);
                if (prev) { candL = dp; cbL = ((uint64_t)dhi << 32) | dlo; }
            }
            if (loseS) {
                grpS = lane_groups(hs, loseS, liveMask);
                unsigned long long const prev = grpS & below_mask((int)lane);
                uint32_t const pd = prev ? 63u - (uint32_t)__clzll((long long)prev) : lane;
                uint32_t const dp = __shfl(p, (int)pd), dlo = __shfl((uint32_t)bytes, (int)pd);
                if (prev) { candS = dp; cbS = dlo; }
            }
            bool const hitL = candL != 0 && candL >= prefixLow && cbL == bytes;                    // :203 MEM_read64 equal
            bool const hitS = candS != 0 && candS >= prefixLow && cbS == (uint32_t)bytes;          // :218 MEM_read32 equal
            bool const hitR = off1 > 0 && rv == (uint32_t)(bytes >> 8);      // :190 repcode at ip+1
            unsigned long long const mL = __ballot(hitL), mR = __ballot(hitR) & searchMask, mS = __ballot(hitS) & searchMask;
            unsigned long long const mAny = (mL & searchMask) | mR | mS;
            int const jE = mAny ? first_lane(mAny) : 64;
            int const Lcommit = jE < 64 ? jE + 1 : K;
            if (jE < 64) evKind = ((mR >> jE) & 1) ? 1 : (((mL >> jE) & 1) ? 2 : 3);
            // :187 hashLong[hl0] = hashSmall[hs0] = curr for every position up to the event: last lane of a group wins
            {   bool const inC = (int)lane < Lcommit;
                unsigned long long const cm = below_mask(Lcommit) & ~below_mask((int)lane + 1);
                if (inC && (grpL & cm) == 0) tabL[hl] = DF_ENTRY(p, tgL);
                if (inC && (grpS & cm) == 0) tabS[hs] = DF_ENTRY(p, tgS);
            }
            __builtin_amdgcn_wave_barrier();
#ifdef ZHIP_DBG_PRINT
            if (lane == 0) printf("  dfbatch ip=%u step=%u K=%d jE=%d kind=%d mL=%llx mS=%llx mR=%llx\n", ip, step, K, jE, evKind, mL, mS, mR);
#endif
            if (evKind) {
                evAvg16 = (3 * evAvg16 + (((uint32_t)jE + 1) << 4)) >> 2;
                kCap = ((evAvg16 * kMul) >> 7) + kAdd; if (kCap > 63) kCap = 63; if (kCap < 2) kCap = 2;
                curr = __builtin_amdgcn_readlane(p, jE);
                candE = __builtin_amdgcn_readlane(evKind == 2 ? candL : candS, jE);
                ip1 = __builtin_amdgcn_readlane(p, jE + 1);                  // lane jE+1 <= K is live
                cand1 = __builtin_amdgcn_readlane(candL, jE + 1);
                long1 = ((mL >> (jE + 1)) & 1) && cand1 > prefixLow;         // :260 long match at ip1 (8 bytes equal, index > lowest)
                if (evKind != 1 && step < 4) {                               // :283-291 hashLong[hl1] = ip1
                    if ((int)lane == jE + 1) tabL[hl] = DF_ENTRY(p, tgL);
                    __builtin_amdgcn_wave_barrier();
                }
                break;
            }
            ZWPROF_COUNT(out, 12, 1);
            ip = ip + (uint32_t)K * step;                                    // :236-237
            kCap = kCap * 2 > 63 ? 63 : kCap * 2;
            if (lastEnd) break;
            if (lastInc) { step++; nextStep += 256; }
        }
        ZWPROF_SYNC(out, 7);
        if (evKind == 0) break;

        uint32_t mLength, offBase;
        uint32_t mstart = curr;
        if (evKind == 1) {                                                   // :190-195
            mstart = curr + 1;
            DfExt const e = df_extend(src, nm8, mstart + 4, mstart + 4 - off1, false, 0, 0, false, 0, 0, 0, 0, 0, 0);
            mLength = 4 + e.fwdM;
            offBase = 1;
        } else {
            uint32_t match = candE;
            uint32_t const k0 = evKind == 2 ? 8u : 4u;                       // :203-209 / :248-264 _search_next_long
            bool const useL = evKind == 3 && long1;
            // catch-up limits (:207, :267) of both pairs: literals available and distance of the candidate to the window's low end
            uint32_t const limM = (curr - anchor) < candE - prefixLow ? (curr - anchor) : candE - prefixLow;
            uint32_t const limL = useL ? ((ip1 - anchor) < cand1 - prefixLow ? (ip1 - anchor) : cand1 - prefixLow) : 0;
            DfExt const e = df_extend(src, nm8, curr + k0, match + k0, useL, ip1 + 8, cand1 + 8, true, curr, candE, limM, ip1, cand1, limL);
            mLength = k0 + e.fwdM;
            uint32_t back = e.backM;
            if (useL) {
                uint32_t const l1len = 8 + e.fwdL;
                if (l1len > mLength) { mstart = ip1; mLength = l1len; match = cand1; back = e.backL; }
            }
            uint32_t const offset = mstart - match;
            mstart -= back; mLength += back;
            off2 = off1; off1 = offset;
            offBase = offset + 3;
        }
#ifdef ZHIP_DBG_PRINT
        if (lane == 0) printf("  dfmatch curr=%u kind=%d mstart=%u len=%u offBase=%u\n", curr, evKind, mstart, mLength, offBase);
#endif
        lits_copy(out, src, nm8, anchor, mstart - anchor);
        store_seq(out, mstart - anchor, offBase, mLength);
        ip = mstart + mLength; anchor = ip;
        ZWPROF_SYNC(out, 8);
        }

        have = false;
        if ((int32_t)ip <= ilimit) {                                         // :300-320
            // ONE round of loads per pass (round 3; before: one for the inserts, one per repcode test, one for the next batch): lanes 0..2
            // fetch the bytes of the complementary inserts (first pass only), every lane compares 8 bytes at ip + 8*lane with the bytes
            // off2 back (the immediate repcode and its length), and the source bytes of the batch that starts at ip ride along
            bool first = postFirst;
            for (;;) {
                uint32_t const q = lane == 0 ? curr + 2 : (lane == 1 ? ip - 2 : ip - 1);
                uint64_t b = 0;
                if (first && lane < 3) b = ld64(src + (q < nm8 ? q : nm8));
                uint32_t const fq = ip + 8u * lane;
                uint32_t const same = off2 > 0 ? lane_same_fwd(src, fq, off2, nm8) : 0u;
                uint64_t const ipb = ld64(src + ip);                         // (the bytes at ip: hashed when the repcode is taken)
                uint32_t const pn = ip + lane, pnc = pn < nm8 ? pn : nm8;
                nbytes = ld64(src + pnc); nrv = ld32(src + (pnc + 1 - off1));
                if (first) {
                    // complementary inserts: long[curr+2], long[ip-2], short[curr+2], short[ip-1] — in this order
                    uint32_t const vv = mulhi64_top32(b, 0xCF1BBCDCB7A56463ULL);
                    uint32_t const hL = vv >> shL, hS = hash_pos<MLS>(b, shS);
                    uint32_t const qL = DF_ENTRY(q, df_tag_long(vv)), qS = DF_ENTRY(q, df_tag_short((uint32_t)b));
                    if (lane == 0 && !skipCurr2) { tabL[hL] = qL; tabS[hS] = qS; }   // (a carrying window has stored curr+2 itself)
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 1) tabL[hL] = qL;
                    if (lane == 2) tabS[hS] = qS;
                    __builtin_amdgcn_wave_barrier();
                    first = false;
                }
                unsigned long long const stop = __ballot(same < 8);
                uint32_t rl = 0;
                if (off2 > 0) {
                    if (stop) { int const f = first_lane(stop); rl = 8u * (uint32_t)f + __builtin_amdgcn_readlane(same, f); }
                    else rl = 512 + wave_count_fwd(src, ip + 512, ip + 512 - off2, nm8);
                }
                if (rl < 4) { have = true; break; }                          // no immediate repcode: the next batch's bytes are at hand
                {   uint32_t const t = off2; off2 = off1; off1 = t; }
                if (lane == 0) {
                    uint32_t const vv = mulhi64_top32(ipb, 0xCF1BBCDCB7A56463ULL);
                    tabS[hash_pos<MLS>(ipb, shS)] = DF_ENTRY(ip, df_tag_short((uint32_t)ipb));
                    tabL[vv >> shL] = DF_ENTRY(ip, df_tag_long(vv));
                }
                __builtin_amdgcn_wave_barrier();
                store_seq(out, 0, 1, rl);
                ip += rl; anchor = ip;
                if ((int32_t)ip > ilimit) break;
            }
        }
        ZWPROF_SYNC(out, 9);
    }
    lits_copy(out, src, nm8, anchor, n - anchor);                           // trailing literals
    lits_flush(out);
    } else {
        for (uint32_t i = lane; i < n - b0; i += 64) lits[i] = src[b0 + i];
        out.litPos = n - b0;
    }
    // ---- _cleanup (:248-256)
    saved2 = (saved1 != 0 && off1 != 0) ? saved1 : saved2;
    if (lane == 0) {
        meta->nbSeq = out.nbSeq; meta->lastLits = n - anchor;
        meta->longPos = out.longPos; meta->longType = out.longType;
        meta->rep[0] = off1 ? off1 : saved1; meta->rep[1] = off2 ? off2 : saved2; meta->rep[2] = repIn3;
        meta->status = 0; meta->litSize = out.litPos; meta->pad0 = 0;
    }
    ZPROF_FLUSH(0);
}

// One unit = one block with fresh tables (tagged 17-bit entries)
template <uint32_t MLS>
__device__ inline void parse_dfast_unit(const uint8_t* __restrict__ src, uint32_t n, const ZhipUnit& u, unsigned char* smem,
                                        uint32_t* __restrict__ tabL, uint32_t* __restrict__ tabS,
                                        ZhipSeq* seqs, uint8_t* lits, ZhipParse* meta)
{
    uint32_t const lane = (uint32_t)lane_id();
    {   // fresh tables (zstd_compress.c:2020): the long and the short table are contiguous
        uint32_t const words = (uint32_t)(dfast_table_bytes(u.hashLog, u.chainLog) >> 2);
        uint4 const z = {0, 0, 0, 0};
        for (uint32_t i = 4 * lane; i < words; i += 256) *(uint4*)(tabL + i) = z;      // tables are 16-byte aligned, sizes multiples of 16 words
    }
    __builtin_amdgcn_wave_barrier();
    parse_dfast_block<MLS, false>(src, 0, n, 0, 1, 1, 4, 8, u, smem, tabL, tabS, seqs, lits, meta);   // lowest index 0, ip = 1 -> maxRep = 1
}

}  // namespace zhip
