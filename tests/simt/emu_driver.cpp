// tests/simt/emu_driver.cpp — TEST INFRASTRUCTURE ONLY.
// Runs the product's kernels (zstd_amd/csrc/zhip_kernels.h, unmodified) on the host SIMT emulator so that the
// wave-level logic can be checked against the oracle without a GPU.  Built by tests/_libs.py with g++.
#include <hip/hip_runtime.h>
#ifdef ZHIP_LZ_STATS
namespace zhip { unsigned long long zhip_lz_stats[8]; }
#endif
#include "zhip_kernels.h"
#include "zhip_cdict_host.h"
#include "zhip_ddict_host.h"

#include <vector>
#include <stdlib.h>
#include <stdio.h>
// full-size slots (fixed strides), what the host library fills for 128 KB units
static std::vector<ZhipSlot> fixed_slots(uint32_t nUnits)
{
    std::vector<ZhipSlot> v(nUnits ? nUnits : 1);
    for (uint32_t i = 0; i < nUnits; i++) { v[i].seqOff = (uint64_t)i * ZHIP_SEQ_CAP; v[i].litOff = (uint64_t)i * ZHIP_LIT_STRIDE; v[i].outOff = (uint64_t)i * ZHIP_OUT_STRIDE; v[i].seqCap = ZHIP_SEQ_CAP; v[i].pad0 = 0; }
    return v;
}

extern "C" {

// parse `nUnits` units with the strategy-fast kernel. seqs: nUnits*ZHIP_SEQ_CAP records, metas: nUnits
void emu_parse_fast(const uint8_t* src, const ZhipUnit* units, uint32_t nUnits, ZhipSeq* seqs, uint8_t* lits, ZhipParse* metas,
                    uint32_t smemBytes, int osThreads)
{
    std::vector<ZhipSlot> const sv = fixed_slots(nUnits); const ZhipSlot* const slots = sv.data();
    simt::launch({nUnits, 1, 1}, {64, 1, 1}, smemBytes,
                 [=] { zhip::k_parse_fast(src, units, slots, nUnits, seqs, lits, metas); }, osThreads);
}

// the queue form of the same stage: dispatch order from k_order_cost + k_order_sort, then persistent workgroups on one ticket counter.
// mode 1 = LDS tables (k_parse_fast_q), 2 = tables in global memory (k_parse_fast_g), 3 = both kernels, one after the other on one queue
// k_order_sort's second job: the number of global-table workgroups a batch of these costs gets (0 = all that were launched)
uint32_t emu_order_sort_limit(const uint32_t* cost, uint32_t nUnits, uint32_t gSparse, uint32_t denseCost)
{
    std::vector<uint32_t> order(nUnits + 1); uint32_t lim = 0;
    uint32_t* const po = order.data(); uint32_t* const pl = &lim;
    simt::launch({1, 1, 1}, {1024, 1, 1}, 0, [=] { zhip::k_order_sort(cost, nUnits, po, pl, gSparse, denseCost); }, 1);
    return lim;
}
void emu_parse_fast_queue(const uint8_t* src, const ZhipUnit* units, uint32_t nUnits, ZhipSeq* seqs, uint8_t* lits, ZhipParse* metas,
                          uint32_t smemBytes, int mode, uint32_t* orderOut, int osThreads)
{
    std::vector<ZhipSlot> const sv = fixed_slots(nUnits); const ZhipSlot* const slots = sv.data();
    std::vector<uint32_t> cost(nUnits + 1), order(nUnits + 1), queue(16, 0);
    uint32_t* const pc = cost.data(); uint32_t* const po = order.data(); uint32_t* const pq = queue.data();
    simt::launch({nUnits, 1, 1}, {64, 1, 1}, 0, [=] { zhip::k_order_cost(src, units, nUnits, pc); }, osThreads);
    simt::launch({1, 1, 1}, {1024, 1, 1}, 0, [=] { zhip::k_order_sort(pc, nUnits, po, nullptr, 0, 0); }, 1);
    if (orderOut) for (uint32_t i = 0; i < nUnits; i++) orderOut[i] = po[i];
    uint32_t maxH = 6; for (uint32_t i = 0; i < nUnits; i++) if (units[i].hashLog > maxH) maxH = units[i].hashLog;
    uint32_t const gw = 1u << maxH, gridG = 3, gridQ = 2;
    std::vector<uint32_t> gt((size_t)gridG * gw, 0xEEEEEEEEu); uint32_t* const pg = gt.data();
    if (mode == 3) {
        // the first kernel's workgroups leave after two units each, the second takes the rest (the queue is shared)
        uint32_t const half = nUnits / 2;
        simt::launch({gridQ, 1, 1}, {64, 1, 1}, smemBytes, [=] { zhip::k_parse_fast_q(src, units, slots, half, seqs, lits, metas, po, pq); }, osThreads);
        pq[0] = half;
        simt::launch({gridG, 1, 1}, {64, 1, 1}, 0, [=] { zhip::k_parse_fast_g(src, units, slots, nUnits, seqs, lits, metas, po, pq, pg, gw); }, osThreads);
    } else if (mode == 2 || mode == 4) {
        if (mode == 4) pq[1] = 1;              // k_order_sort's decision: one global-table workgroup takes part, the other two leave at once
        simt::launch({gridG, 1, 1}, {64, 1, 1}, 0, [=] { zhip::k_parse_fast_g(src, units, slots, nUnits, seqs, lits, metas, po, pq, pg, gw); }, osThreads);
    } else {
        simt::launch({gridQ, 1, 1}, {64, 1, 1}, smemBytes, [=] { zhip::k_parse_fast_q(src, units, slots, nUnits, seqs, lits, metas, po, pq); }, osThreads);
    }
}

void emu_parse_dfast(const uint8_t* src, const ZhipUnit* units, uint32_t nUnits, uint32_t* tabs, size_t tabStride,
                     ZhipSeq* seqs, uint8_t* lits, ZhipParse* metas, int osThreads)
{
    std::vector<ZhipSlot> const sv = fixed_slots(nUnits); const ZhipSlot* const slots = sv.data();
    simt::launch({nUnits, 1, 1}, {64, 1, 1}, zhip::dfast_lds_bytes(),
                 [=] { zhip::k_parse_dfast(src, units, slots, nUnits, tabs, tabStride, seqs, lits, metas, nullptr); }, osThreads);
}
uint64_t emu_dfast_table_bytes(uint32_t hashLog, uint32_t chainLog) { return zhip::dfast_table_bytes(hashLog, chainLog); }


// hash-chain strategies (greedy / lazy / lazy2): the three launches of zhip_parse_lazy.h
void emu_parse_lazy(const uint8_t* src, const ZhipUnit* units, uint32_t nUnits, uint32_t* tabs, size_t tabStride, uint64_t* best,
                    ZhipSeq* seqs, uint8_t* lits, ZhipParse* metas, int osThreads)
{
    std::vector<ZhipSlot> const sv = fixed_slots(nUnits); const ZhipSlot* const slots = sv.data();
    uint32_t maxLen = 1, maxHlog = 6;
    for (uint32_t i = 0; i < nUnits; i++) { if (units[i].srcLen > maxLen) maxLen = units[i].srcLen; if (units[i].hashLog > maxHlog) maxHlog = units[i].hashLog; }
    simt::launch({nUnits, 1, 1}, {64, 1, 1}, zhip::hc_chain_lds_bytes(maxHlog),
                 [=] { zhip::k_hc_chain(src, units, nUnits, tabs, tabStride, best); }, osThreads);
    uint32_t const bpu = (maxLen + ZHIP_HC_SEARCH_THREADS - 1) / ZHIP_HC_SEARCH_THREADS;
    if (getenv("ZHIP_EMU_HC_GLOBAL"))          // the L2-resident variant (kept for comparison runs)
        simt::launch({((nUnits + 7) / 8) * 8 * bpu, 1, 1}, {ZHIP_HC_SEARCH_THREADS, 1, 1}, 0,
                     [=] { zhip::k_hc_search(src, units, nUnits, bpu, tabs, tabStride, best); }, osThreads);
    else
        simt::launch({nUnits, 1, 1}, {ZHIP_HC_SEARCH_LDS_THREADS, 1, 1}, ((maxLen + 15) & ~15u) + 32,
                     [=] { zhip::k_hc_search_lds(src, units, nUnits, tabs, tabStride, best, (const ZhipParse*)nullptr); }, osThreads);
    // the row matcher's two-pass prediction, as the host library launches it ($ZHIP_RH_PREDICT=1 turns it on, as in the library; $ZHIP_RH_BUDGET: live searches a TRY parse may make)
    bool anyRow = false; for (uint32_t i = 0; i < nUnits; i++) anyRow = anyRow || units[i].rowLog != 0;
    const char* const pe = getenv("ZHIP_RH_PREDICT"); const char* const be = getenv("ZHIP_RH_BUDGET");
    uint32_t const budget = be ? (uint32_t)atoi(be) : 256u;
    const ZhipParse* const cm = metas;
    if (anyRow && pe && atoi(pe) != 0) {
        simt::launch({nUnits, 1, 1}, {64, 1, 1}, ZHIP_RH_DIRTY_BYTES,
                     [=] { zhip::k_parse_lazy(src, units, slots, nUnits, tabs, tabStride, best, seqs, lits, metas, 2u, budget); }, osThreads);
        simt::launch({nUnits, 1, 1}, {64, 1, 1}, ZHIP_RH_DIRTY_BYTES,
                     [=] { zhip::k_parse_lazy(src, units, slots, nUnits, tabs, tabStride, best, seqs, lits, metas, 1u, 0u); }, osThreads);
        simt::launch({nUnits, 1, 1}, {ZHIP_HC_SEARCH_LDS_THREADS, 1, 1}, ((maxLen + 15) & ~15u) + 32,
                     [=] { zhip::k_hc_search_lds(src, units, nUnits, tabs, tabStride, best, cm); }, osThreads);
        simt::launch({nUnits, 1, 1}, {64, 1, 1}, ZHIP_RH_DIRTY_BYTES,
                     [=] { zhip::k_parse_lazy(src, units, slots, nUnits, tabs, tabStride, best, seqs, lits, metas, 3u, 0u); }, osThreads);
    } else
    simt::launch({nUnits, 1, 1}, {64, 1, 1}, ZHIP_RH_DIRTY_BYTES,
                 [=] { zhip::k_parse_lazy(src, units, slots, nUnits, tabs, tabStride, best, seqs, lits, metas, 0u, 0u); }, osThreads);
}
uint64_t emu_hc_table_words(uint32_t hashLog) { return zhip::hc_table_words(hashLog); }

// dictionary path: build the CDict with the product's host code, fill the records' working parameters, parse them with
// k_parse_dict.  Records lie back to back in src (offsets[nRec+1]); slots are the fixed full-size ones (test harness).
// returns 0 ok, >0 = host_cdict_build error, -1 = some record is above the attach cut-off
int emu_parse_dict(const uint8_t* src, const uint64_t* offsets, uint32_t nRec, const uint8_t* dict, size_t dictSize, int level,
                   ZhipUnit* unitsOut, ZhipSeq* seqs, uint8_t* lits, ZhipParse* metas, int osThreads)
{
    zhip::HostCDict cd;
    int const e = zhip::host_cdict_build(cd, dict, dictSize, level);
    if (e) return e;
    if (cd.len == 0) return 9;
    uint32_t mh = 6, mc = 6;
    std::vector<uint32_t> extIdx;
    for (uint32_t i = 0; i < nRec; i++) {
        zhip::CParams cp; size_t const n = (size_t)(offsets[i + 1] - offsets[i]);
        bool const copyMode = zhip::host_cdict_is_copy_mode(cd, n);
        if (!(copyMode ? zhip::host_cdict_copy_params(cd, n, &cp) : zhip::host_cdict_unit_params(cd, n, &cp))) return -1;
        ZhipUnit& u = unitsOut[i];
        u.srcOff = offsets[i]; u.srcLen = (uint32_t)n; u.windowLog = (uint8_t)cp.windowLog; u.chainLog = (uint8_t)cp.chainLog; u.hashLog = (uint8_t)cp.hashLog;
        u.minMatch = (uint8_t)cp.minMatch; u.strategy = (uint8_t)cp.strategy; u.searchLog = (uint8_t)cp.searchLog; u.litMode = 0; u.pad0 = copyMode ? ZHIP_UNIT_COPYMODE : 0; u.targetLength = cp.targetLength;
        if (copyMode) { extIdx.push_back(i); continue; }
        if (cp.hashLog > mh) mh = cp.hashLog;
        if (cp.chainLog > mc) mc = cp.chainLog;
    }
    zhip::ZhipCDictDev dv;
    dv.content = cd.content.data(); dv.len = (uint32_t)cd.len; dv.hashLog = cd.cp.hashLog; dv.chainLog = cd.cp.chainLog; dv.minMatch = cd.cp.minMatch;
    dv.strategy = cd.cp.strategy; dv.tabL = cd.tabL.data(); dv.tabS = cd.tabS.data(); dv.rep[0] = cd.rep[0]; dv.rep[1] = cd.rep[1]; dv.rep[2] = cd.rep[2]; dv.dictID = cd.dictID;
    std::vector<ZhipSlot> const sv = fixed_slots(nRec); const ZhipSlot* const slots = sv.data();
    const ZhipUnit* units = unitsOut;
    int const qmode = getenv("ZHIP_EMU_DICT_QUEUE") ? atoi(getenv("ZHIP_EMU_DICT_QUEUE")) : 0;   // 1: k_parse_dict_q, 2: k_parse_dict_g (tables in global memory)
    uint32_t const ldsB = cd.cp.strategy == 1 ? zhip::dict_fast_lds_bytes(mh) : zhip::dict_lds_bytes(mh, mc);
    if (extIdx.size() < nRec && qmode) {
        std::vector<uint32_t> queue(16, 0); uint32_t* const pq = queue.data();
        uint32_t const grid = 3, gtabBytes = ((2u << mh) + (2u << mc) + 255u) & ~255u;
        std::vector<unsigned char> gt((size_t)grid * gtabBytes, 0xEE); unsigned char* const pg = gt.data();
        if (qmode == 2)
            simt::launch({grid, 1, 1}, {64, 1, 1}, 0, [=] { zhip::k_parse_dict_g(src, units, slots, nRec, dv, seqs, lits, metas, pq, pg, gtabBytes); }, osThreads);
        else
            simt::launch({grid, 1, 1}, {64, 1, 1}, ldsB, [=] { zhip::k_parse_dict_q(src, units, slots, nRec, dv, seqs, lits, metas, pq); }, osThreads);
    } else
    if (extIdx.size() < nRec)
        simt::launch({nRec, 1, 1}, {64, 1, 1}, ldsB,
                     [=] { zhip::k_parse_dict(src, units, slots, nRec, dv, seqs, lits, metas); }, osThreads);
    if (!extIdx.empty()) {                                     // copy mode (sources above the attach cut-off), like the host library launches it
        uint32_t const nExt = (uint32_t)extIdx.size();
        size_t stride = zhip::ext_table_words(cd.cp.hashLog, cd.cp.chainLog, cd.cp.strategy); stride = (stride + 3) & ~(size_t)3;
        std::vector<uint32_t> tabs((size_t)nExt * stride, 0xEEEEEEEEu);
        uint32_t* const tp = tabs.data(); const uint32_t* const ip = extIdx.data();
        uint32_t const wordsL = 1u << cd.cp.hashLog, wordsS = cd.cp.strategy == 2 ? 1u << cd.cp.chainLog : 0u;
        const uint32_t* const tl = cd.tabL.data(); const uint32_t* const ts = cd.tabS.data();
        simt::launch({8, nExt, 1}, {256, 1, 1}, 0, [=] { zhip::k_ext_init(tl, ts, wordsL, wordsS, tp, stride); }, osThreads);
        simt::launch({(nExt + 63) / 64, 1, 1}, {64, 1, 1}, 0,
                     [=] { zhip::k_parse_ext(src, units, slots, ip, nExt, dv, tp, stride, seqs, lits, metas); }, osThreads);
    }
    return 0;
}

// decoder: frames[] describe where each frame lies in src and where its content goes in dst; a dictionary is optional.
// returns 0, or the host_ddict_build error
int emu_decode(const uint8_t* src, const ZhipDFrame* frames, uint32_t nFrames, uint8_t* dst, const uint8_t* dict, size_t dictSize,
               ZhipDResult* results, uint32_t nGroups, int osThreads)
{
    zhip::HostDDict dd; ZhipDDictDev dv; memset(&dv, 0, sizeof(dv));
    if (dict) {
        int const e = zhip::host_ddict_build(dd, dict, dictSize);
        if (e) return e;
        dv.len = (uint32_t)dd.content.size();
        if (dd.content.empty()) dd.content.push_back(0);
        dv.content = dd.content.data();
        dv.dictID = dd.dictID; dv.hasEntropy = dd.hasEntropy; dv.hufLog = dd.hufLog; dv.huf = dd.huf.data(); dv.huf2 = dd.huf2.data(); dv.fse = dd.fse.data();
        for (int k = 0; k < 3; k++) { dv.log[k] = dd.log[k]; dv.rep[k] = dd.rep[k]; }
    }
    uint64_t defTabs[160]; zhip::host_dec_default_tables(defTabs);
    if (nGroups == 0 || nGroups > nFrames) nGroups = nFrames ? nFrames : 1;
    std::vector<uint8_t> lit((size_t)nGroups * ZHIP_DEC_LIT_STRIDE, 0xEE);
    std::vector<ZhipDSeq> recs((size_t)nGroups * 2 * (ZHIP_DEC_CHUNK + 1));
    uint32_t counter = 0; uint32_t* const cp = &counter;
    uint8_t* const lp = lit.data(); ZhipDSeq* const rp = recs.data(); const uint64_t* const dt = defTabs;
    simt::launch({nGroups, 1, 1}, {ZHIP_DEC_THREADS, 1, 1}, sizeof(zhip::DecShared),
                 [=] { zhip::k_decode(src, frames, nFrames, dst, lp, rp, cp, dv, dt, results); }, osThreads);
    return 0;
}
// ONE frame through the block-parallel decoder (zhip_decode_big.h), the launches of the host library in order.
// returns 0 and *outSize, or the path's status (the caller would then fall back to k_decode); rounds (optional) = jump rounds made
static uint32_t* g_bf_map_out = nullptr;      // analysis hook (tests/tools/jump_rounds.py): receives the copy map as k_bf_build leaves it
void emu_decode_big_want_map(uint32_t* mapOut) { g_bf_map_out = mapOut; }
uint32_t emu_decode_big(const uint8_t* src, uint32_t srcLen, uint8_t* dst, uint32_t dstCap, uint32_t* outSize, uint32_t* checksumOut, uint32_t* rounds, int osThreads)
{
    zhip::BfHeader const H = zhip::bf_parse_header(src, srcLen);
    if (!H.ok) return ZHIP_DE_UNSUPPORTED;
    uint64_t const limit = H.known ? H.fcs : (uint64_t)dstCap;          // as bigframe_decode (zhip_declib.h)
    if (limit > dstCap) return ZHIP_DE_UNSUPPORTED;
    uint32_t const capBlocks = (uint32_t)(limit / 4096 + 4096);
    std::vector<ZhipBfBlock> blocks(capBlocks);
    ZhipBfInfo info; memset(&info, 0, sizeof(info));
    ZhipBfBlock* const bp = blocks.data(); ZhipBfInfo* const ip = &info;
    uint64_t defTabs[160]; zhip::host_dec_default_tables(defTabs); const uint64_t* const dt = defTabs;
    uint32_t const hdrSize = H.hdrSize, blockMax = H.blockMax, hasCk = H.hasChecksum;
    bool const tr = getenv("ZHIP_EMU_TRACE") != nullptr;
#define BFTR(x) do { if (tr) fprintf(stderr, "bf: %s\n", x); } while (0)
    BFTR("walk");
    simt::launch({1, 1, 1}, {64, 1, 1}, 0, [=] { zhip::k_bf_walk(src, srcLen, hdrSize, blockMax, hasCk, bp, capBlocks, ip); }, 1);
    if (info.status) return info.status;
    uint32_t const nB = info.nBlocks;
    BFTR("prep");
    simt::launch({(nB + 255) / 256, 1, 1}, {256, 1, 1}, 0, [=] { zhip::k_bf_prep(src, blockMax, bp, ip); }, osThreads);
    BFTR("deps");
    simt::launch({1, 1, 1}, {64, 1, 1}, 0, [=] { zhip::k_bf_deps(bp, ip); }, 1);
    if (info.status) return info.status;
    if (info.totalRecs > limit + nB) return ZHIP_DE_UNSUPPORTED;
    // arenas with 256 canary bytes on both sides: a kernel that writes outside what the host sized is reported (0xBAD)
    size_t const CAN = 256;
    std::vector<uint8_t> litV(info.totalLit + 64 + 2 * CAN, 0xA5); std::vector<uint8_t> recV((info.totalRecs + 1) * sizeof(ZhipDSeq) + 2 * CAN, 0xA5);
    uint8_t* const lp = litV.data() + CAN; ZhipDSeq* const rp = (ZhipDSeq*)(recV.data() + CAN);
    auto canaries_ok = [&](const std::vector<uint8_t>& v) { for (size_t i = 0; i < CAN; i++) if (v[i] != 0xA5 || v[v.size() - 1 - i] != 0xA5) return false; return true; };
    BFTR("entropy");
    simt::launch({nB, 1, 1}, {ZHIP_BF_THREADS, 1, 1}, sizeof(zhip::DecShared), [=] { zhip::k_bf_entropy(src, blockMax, bp, ip, lp, rp, dt); }, osThreads);
    BFTR("scan");
    uint32_t const fcs32 = (uint32_t)limit;
    simt::launch({1, 1, 1}, {64, 1, 1}, 0, [=] { zhip::k_bf_scan(bp, ip, fcs32); }, 1);
    if (info.status) return info.status;
    if (H.known && info.totalOut != H.fcs) return ZHIP_DE_CORRUPT;
    uint32_t const n = (uint32_t)info.totalOut;
    if (!canaries_ok(litV) || !canaries_ok(recV)) return 0xBAD;
    std::vector<uint8_t> mapV(((size_t)n + 8) * 4 + 2 * CAN, 0xA5); uint32_t* const mp = (uint32_t*)(mapV.data() + CAN);
    BFTR("build");
    simt::launch({nB, 1, 1}, {256, 1, 1}, 0, [=] { zhip::k_bf_build(src, bp, ip, lp, rp, dst, mp); }, osThreads);
    if (!canaries_ok(mapV)) return 0xBAD;
    if (info.status) return info.status;
    if (g_bf_map_out) memcpy(g_bf_map_out, mp, (size_t)n * 4);
    uint32_t r = 0;
    if (n) for (; r < 64; r++) {
        info.changed = 0;
        simt::launch({(n + 1023) / 1024, 1, 1}, {256, 1, 1}, 0, [=] { zhip::k_bf_jump(mp, n, ip); }, osThreads);
        if (!info.changed) break;
    }
    if (rounds) *rounds = r;
    if (n) simt::launch({(n + 1023) / 1024, 1, 1}, {256, 1, 1}, 0, [=] { zhip::k_bf_copy(mp, dst, n); }, osThreads);
    if (!canaries_ok(litV) || !canaries_ok(recV) || !canaries_ok(mapV)) return 0xBAD;
    *outSize = n; if (checksumOut) *checksumOut = info.checksum;
    return 0;
}
uint32_t emu_sizeof_dframe(void) { return sizeof(ZhipDFrame); }
uint32_t emu_dec_shared(void) { return (uint32_t)sizeof(zhip::DecShared); }

// stage 2 for `nUnits` units: out slots of ZHIP_OUT_STRIDE bytes, outSize[nUnits]
// XXH64 of every unit (frame checksums)
void emu_xxh64(const uint8_t* src, const ZhipUnit* units, uint32_t nUnits, uint32_t* checks, int osThreads)
{
    simt::launch({(nUnits + 15) / 16, 1, 1}, {64, 1, 1}, 0, [=] { zhip::k_xxh64(src, units, nUnits, checks); }, osThreads);
}

// the wave-per-unit variant for large units (frames)
void emu_xxh64_wave(const uint8_t* src, const ZhipUnit* units, uint32_t nUnits, uint32_t* checks, int osThreads)
{
    simt::launch({nUnits, 1, 1}, {64, 1, 1}, ZHIP_XXH_WAVE_LDS, [=] { zhip::k_xxh64_wave(src, units, nUnits, checks); }, osThreads);
}

void emu_entropy_ck(const uint8_t* src, const ZhipUnit* units, uint32_t nUnits, const ZhipSeq* seqs, const ZhipParse* metas,
                    const uint8_t* lits, uint16_t* stBits, uint8_t* out, uint32_t* outSize, const uint32_t* checks, int osThreads);
void emu_entropy(const uint8_t* src, const ZhipUnit* units, uint32_t nUnits, const ZhipSeq* seqs, const ZhipParse* metas,
                 const uint8_t* lits, uint16_t* stBits, uint8_t* out, uint32_t* outSize, int osThreads)
{
    emu_entropy_ck(src, units, nUnits, seqs, metas, lits, stBits, out, outSize, nullptr, osThreads);
}
void emu_entropy_ck(const uint8_t* src, const ZhipUnit* units, uint32_t nUnits, const ZhipSeq* seqs, const ZhipParse* metas,
                    const uint8_t* lits, uint16_t* stBits, uint8_t* out, uint32_t* outSize, const uint32_t* checks, int osThreads)
{
    std::vector<ZhipSlot> const sv = fixed_slots(nUnits); const ZhipSlot* const slots = sv.data();
    // both shapes of the encoder, as the host library launches them: 256 threads for the units above ZHIP_ENT_SMALL_MAX, one wavefront below
    simt::launch({nUnits, 1, 1}, {ZHIP_ENT_THREADS, 1, 1}, sizeof(zhip::EntShared),
                 [=] { zhip::k_entropy(src, units, slots, nUnits, seqs, metas, lits, stBits, out, outSize, nullptr, 0u, checks, 1u); }, osThreads);
    simt::launch({nUnits, 1, 1}, {64, 1, 1}, sizeof(zhip::EntSharedSmall),
                 [=] { zhip::k_entropy_small(src, units, slots, nUnits, seqs, metas, lits, stBits, out, outSize, nullptr, 0u, checks); }, osThreads);
}
// multi-block frames: frames[i] describes one whole input; out holds outStride bytes per frame
void emu_frame_fast(const uint8_t* src, const ZhipUnit* frames, uint32_t nFrames, uint8_t* out, uint64_t outStride, uint32_t* outSize,
                    const uint32_t* checks, int osThreads)
{
    std::vector<ZhipSlot> sv(nFrames ? nFrames : 1);
    uint32_t maxLog = 0;
    for (uint32_t i = 0; i < nFrames; i++) {
        sv[i].seqOff = (uint64_t)i * ZHIP_SEQ_CAP; sv[i].litOff = (uint64_t)i * ZHIP_LIT_STRIDE; sv[i].outOff = (uint64_t)i * outStride; sv[i].seqCap = ZHIP_SEQ_CAP; sv[i].pad0 = 0;
        if (frames[i].hashLog > maxLog) maxLog = frames[i].hashLog;
    }
    const ZhipSlot* const slots = sv.data();
    std::vector<ZhipSeq> seqs((size_t)nFrames * ZHIP_SEQ_CAP); std::vector<uint8_t> lits((size_t)nFrames * ZHIP_LIT_STRIDE);
    std::vector<uint16_t> stBits((size_t)nFrames * ZHIP_SEQ_CAP * 3);
    size_t tabStride = 0; uint32_t ldsTab = 0;
    for (uint32_t i = 0; i < nFrames; i++) {
        uint32_t const mode = zhip::frame_table_mode(frames[i].strategy, frames[i].hashLog, frames[i].srcLen);
        if (mode != zhip::ZHIP_FT_HBM) { uint32_t const b = zhip::frame_table_lds_bytes(mode, frames[i].hashLog); if (b > ldsTab) ldsTab = b; }
        else { size_t const w = zhip::frame_table_words(frames[i].strategy, frames[i].hashLog, frames[i].chainLog); if (w > tabStride) tabStride = w; }
    }
    std::vector<uint32_t> tabs((size_t)nFrames * tabStride + 1);
    std::vector<zhip::ZhipFrameState> states(nFrames ? nFrames : 1);
    ZhipSeq* const sq = seqs.data(); uint8_t* const lt = lits.data(); uint16_t* const sb = stBits.data(); uint32_t* const tb = tabs.data();
    zhip::ZhipFrameState* const stp = states.data();
    simt::launch({nFrames, 1, 1}, {ZHIP_ENT_THREADS, 1, 1}, zhip::frame_lds_bytes(ldsTab),
                 [=] { zhip::k_frame_fast(src, frames, slots, nFrames, tb, tabStride, sq, lt, sb, out, outSize, stp, checks, (const zhip::ZhipJob*)nullptr); }, osThreads);
}
// one frame as jobs (ZSTD_c_nbWorkers semantics): units[i] / jobs[i] describe job i (units[i].srcOff = the frame start)
void emu_frame_jobs(const uint8_t* src, const ZhipUnit* units, const zhip::ZhipJob* jobs, uint32_t nJobs, uint8_t* out, uint64_t outStride,
                    uint32_t* outSize, const uint32_t* checks, int osThreads)
{
    std::vector<ZhipSlot> sv(nJobs ? nJobs : 1);
    size_t tabStride = 0; uint32_t ldsTab = 0;
    for (uint32_t i = 0; i < nJobs; i++) {
        sv[i].seqOff = (uint64_t)i * ZHIP_SEQ_CAP; sv[i].litOff = (uint64_t)i * ZHIP_LIT_STRIDE; sv[i].outOff = (uint64_t)i * outStride; sv[i].seqCap = ZHIP_SEQ_CAP; sv[i].pad0 = 0;
        uint32_t const mode = zhip::frame_table_mode(units[i].strategy, units[i].hashLog, (uint64_t)units[i].srcLen + jobs[i].prefixLen + 1u);
        if (mode != zhip::ZHIP_FT_HBM) { uint32_t const b = zhip::frame_table_lds_bytes(mode, units[i].hashLog); if (b > ldsTab) ldsTab = b; }
        else { size_t const w = zhip::frame_table_words(units[i].strategy, units[i].hashLog, units[i].chainLog); if (w > tabStride) tabStride = w; }
    }
    const ZhipSlot* const slots = sv.data();
    std::vector<ZhipSeq> seqs((size_t)nJobs * ZHIP_SEQ_CAP); std::vector<uint8_t> lits((size_t)nJobs * ZHIP_LIT_STRIDE);
    std::vector<uint16_t> stBits((size_t)nJobs * ZHIP_SEQ_CAP * 3);
    std::vector<uint32_t> tabs((size_t)nJobs * tabStride + 1);
    std::vector<zhip::ZhipFrameState> states(nJobs ? nJobs : 1);
    ZhipSeq* const sq = seqs.data(); uint8_t* const lt = lits.data(); uint16_t* const sb = stBits.data(); uint32_t* const tb = tabs.data();
    zhip::ZhipFrameState* const stp = states.data();
    simt::launch({nJobs, 1, 1}, {ZHIP_ENT_THREADS, 1, 1}, zhip::frame_lds_bytes(ldsTab),
                 [=] { zhip::k_frame_fast(src, units, slots, nJobs, tb, tabStride, sq, lt, sb, out, outSize, stp, checks, jobs); }, osThreads);
}
// multi-block frames / jobs of the lazy strategies (zhip_frame_lazy.h): units[i] (+ jobs[i], or jobs == nullptr for whole frames);
// the three launches of the host library, in order
void emu_frame_lazy(const uint8_t* src, const ZhipUnit* units, const zhip::ZhipJob* jobs, uint32_t nW, uint8_t* out, uint64_t outStride,
                    uint32_t* outSize, const uint32_t* checks, int osThreads)
{
    std::vector<ZhipSlot> sv(nW ? nW : 1);
    std::vector<zhip::ZhipLzSlot> lv(nW ? nW : 1);
    uint64_t posTotal = 0, headTotal = 0, ringTotal = 0; uint32_t longest = 1;
    for (uint32_t i = 0; i < nW; i++) {
        sv[i].seqOff = (uint64_t)i * ZHIP_SEQ_CAP; sv[i].litOff = (uint64_t)i * ZHIP_LIT_STRIDE; sv[i].outOff = (uint64_t)i * outStride; sv[i].seqCap = ZHIP_SEQ_CAP; sv[i].pad0 = 0;
        zhip::lz_fill_slot(lv[i], units[i], jobs ? jobs[i].prefixLen : 0u, posTotal, headTotal, ringTotal);
        if (units[i].srcLen > longest) longest = units[i].srcLen;
    }
    const ZhipSlot* const slots = sv.data(); const zhip::ZhipLzSlot* const lz = lv.data();
    std::vector<ZhipSeq> seqs((size_t)nW * ZHIP_SEQ_CAP); std::vector<uint8_t> lits((size_t)nW * ZHIP_LIT_STRIDE);
    std::vector<uint16_t> stBits((size_t)nW * ZHIP_SEQ_CAP * 3);
    std::vector<uint32_t> prev(posTotal + 16, 0xDDDDDDDDu), heads(headTotal + 16, 0xDDDDDDDDu);
    std::vector<uint8_t> tags(posTotal + 16, 0xDD);
    const char* const re = getenv("ZHIP_LZ_RING");               // the live rows (as in the library: on unless $ZHIP_LZ_RING=0)
    std::vector<uint8_t> rings((re && atoi(re) == 0) ? 0 : ringTotal + 256, 0xDD);
    uint8_t* const rg = rings.empty() ? (uint8_t*)nullptr : rings.data();
    std::vector<zhip::LzRec> best(posTotal + 16);
    memset(best.data(), 0xDD, best.size() * sizeof(zhip::LzRec));
    std::vector<zhip::ZhipFrameState> states(nW ? nW : 1);
    ZhipSeq* const sq = seqs.data(); uint8_t* const lt = lits.data(); uint16_t* const sb = stBits.data();
    uint32_t* const pv = prev.data(); uint32_t* const hd = heads.data(); uint8_t* const tg = tags.data(); zhip::LzRec* const bs = best.data();
    zhip::ZhipFrameState* const stp = states.data();
    simt::launch({nW, 1, 1}, {ZHIP_LZ_LINK_THREADS, 1, 1}, sizeof(zhip::LzLinkShared),
                 [=] { zhip::k_lz_links(src, units, jobs, lz, nW, pv, tg, hd); }, osThreads);
    simt::launch({(longest + 255) / 256, nW, 1}, {256, 1, 1}, 0,
                 [=] { zhip::k_lz_search(src, units, jobs, lz, 0u, nW, pv, tg, bs, (const zhip::ZhipFrameState*)nullptr); }, osThreads);
    uint32_t havePred = 0;
    {   const char* const pe = getenv("ZHIP_LZ_PREDICT");        // the two-pass prediction, as the host library launches it (on unless $ZHIP_LZ_PREDICT=0, as in the library)
        if (!(pe && atoi(pe) == 0)) {
            havePred = 1;
            simt::launch({nW, 1, 1}, {64, 1, 1}, sizeof(ZhipParse), [=] { zhip::k_lz_predict(src, units, jobs, lz, nW, pv, tg, bs, stp); }, osThreads);
            simt::launch({(longest + 255) / 256, nW, 1}, {256, 1, 1}, 0, [=] { zhip::k_lz_search(src, units, jobs, lz, 0u, nW, pv, tg, bs, (const zhip::ZhipFrameState*)stp); }, osThreads);
        }
    }
    simt::launch({nW, 1, 1}, {ZHIP_ENT_THREADS, 1, 1}, zhip::frame_lazy_lds_bytes(),
                 [=] { zhip::k_frame_lazy(src, units, slots, jobs, lz, nW, pv, tg, bs, hd, rg, sq, lt, sb, out, outSize, stp, checks, havePred); }, osThreads);
}
#ifdef ZHIP_LZ_STATS
void emu_lz_stats(unsigned long long* out, int reset) { for (int i = 0; i < 8; i++) { out[i] = zhip::zhip_lz_stats[i]; if (reset) zhip::zhip_lz_stats[i] = 0; } }
#endif
uint32_t emu_sizeof_job(void) { return (uint32_t)sizeof(zhip::ZhipJob); }
uint32_t emu_out_stride(void) { return ZHIP_OUT_STRIDE; }
uint32_t emu_lit_stride(void) { return ZHIP_LIT_STRIDE; }
uint32_t emu_ent_shared(void) { return (uint32_t)sizeof(zhip::EntShared); }

uint32_t emu_fast_lds_bytes(uint32_t hlog) { return zhip::fast_tag_lds_bytes(hlog); }
uint32_t emu_seq_cap(void) { return ZHIP_SEQ_CAP; }
uint32_t emu_sizeof_unit(void) { return sizeof(ZhipUnit); }
uint32_t emu_sizeof_parse(void) { return sizeof(ZhipParse); }


// stage hooks: the wave-wide table builders on caller-supplied histograms (one case per workgroup)
void emu_test_huf(const uint32_t* counts, const uint32_t* maxSyms, uint32_t nCases, uint32_t maxNbBits, uint32_t* codes, uint8_t* hdrs, uint32_t* meta)
{
    simt::launch({nCases, 1, 1}, {64, 1, 1}, sizeof(zhip::ZhipTestHufShared), [=] { zhip::k_test_huf(counts, maxSyms, maxNbBits, codes, hdrs, meta); }, 0);
}
void emu_test_fse(const uint32_t* counts, const uint32_t* params, uint32_t nCases, int16_t* norms, uint8_t* ncounts, int32_t* meta, zhip::FseCTable* tables)
{
    simt::launch({nCases, 1, 1}, {64, 1, 1}, sizeof(zhip::ZhipTestFseShared), [=] { zhip::k_test_fse(counts, params, norms, ncounts, meta, tables); }, 0);
}
uint32_t emu_sizeof_fse_ctable(void) { return (uint32_t)sizeof(zhip::FseCTable); }

}
