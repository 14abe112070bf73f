#!/usr/bin/env python3
"""scripts/isa_phase_table.py [--mls N] [--json] [--hits FILE] — per-phase instruction table of the ZSTD_fast window (zhip_parse.h window_batch),
read from the gfx950 assembly.  No GPU needed.

The parser is compiled once (one instantiation, the product's code generation options) with -DZHIP_WPH_MARK: every ZWPH(out, id) marker of
zhip_parse.h then leaves a "; ZWPH id" comment in the assembly and nothing else.  A marker CLOSES phase id, so an instruction belongs to the
phase of the next marker control flow reaches from it: a backward data-flow pass over the kernel's basic blocks (a block whose successors
disagree is reported as "a|b").  Per phase the script prints the static instruction count by class (VALU, cross-lane, SALU, LDS, VMEM, waits,
branches).  With --hits (the visits per unit that scripts/prof_phases.py measures on the GPU with the same markers) the straight-line phases'
counts are multiplied out to wave-instructions per unit and per source byte: the instruction budget of the window the round-5 verdict asked for."""
import argparse, collections, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zstd_amd", "csrc")
NAMES = {0: "F_SRC  pending literals, source + repcode bytes, hash, tag", 1: "F_TAB  table gather (LDS), candidate address", 2: "F_DUP  candidate load issue, mark / peek / unmark",
         3: "F_GRP  hash groups (readlane / ballot loop), p1 / p2", 4: "F_MASK candidate bytes: M, E1 / E2 ballots", 5: "SEARCH one search of the span (per loop turn)",
         6: "M_ELOAD match: inserts, E load of a new offset, ballots", 7: "M_RUNS backward + forward runs (far calls not counted)", 8: "M_EMIT emit, coverage, :407-408, immediate repcodes",
         9: "LEAVE  a match left the window (carry / by loads)", 10: "E_PRE  next window's source bytes requested", 11: "E_TAB  table writes", 12: "E_OUT  sequences + literals stored",
         13: "B_SCAN schedule-shaped batch to its event", 14: "B_MATCH its match (wave_extend, literals, sequence)", 15: "B_POST post_match round(s)", 16: "TAIL", 17: "INIT", 18: "LOOP   between windows (parse_fast_block)",
         19: "CARRY  carried match: insert, immediate-repcode test", 20: "IMM    one immediate repcode (:410-420 from the masks)", 21: "GRP_IT one hash group (readlane + ballot)",
         22: "GRP_NF window with groups: p1 / p2 / m1 / m2", 23: "LATE   one late (NF) insert", 24: "LEAVE_FAR :403-420 by loads (call not counted)"}

KERNEL = r'''
#include <hip/hip_runtime.h>
#include "zhip_common.h"
#include "zhip_kernel_params.h"
#include "zhip_parse.h"
namespace zhip {
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3)))
k_one(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipSlot* __restrict__ slots, uint32_t nUnits,
      ZhipSeq* __restrict__ seqs, uint8_t* __restrict__ lits, ZhipParse* __restrict__ metas, const uint32_t* __restrict__ order, uint32_t* __restrict__ queue)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    for (;;) {
        uint32_t t = 0;
        if ((threadIdx.x & 63) == 0) t = atomicAdd(queue, 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= nUnits) return;
        uint32_t const ui = order ? order[t] : t;
        ZhipUnit const u = units[ui];
        ZhipSlot const sl = slots[ui];
        parse_fast_unit<%(mls)d>(src + u.srcOff, u.srcLen, u, smem, seqs + sl.seqOff, lits + sl.litOff, metas + ui);
        __builtin_amdgcn_wave_barrier();
    }
}
}
'''
LISTING = []
COLS = ["valu", "xlane", "salu", "lds", "vmem", "smem", "wait", "branch", "nop"]


def classify(ins):
    op = ins.split()[0]
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_getpc", "s_endpgm")): return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_store")): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")): return "xlane"
    if op.startswith("v_"): return "valu"
    return "nop"


def compile_asm(mls, defs):
    tmp = tempfile.mkdtemp()
    srcp, asm = os.path.join(tmp, "one.hip"), os.path.join(tmp, "one.s")
    open(srcp, "w").write(KERNEL % dict(mls=mls))
    sys.path.insert(0, ROOT)
    from zstd_amd.build import UNIT_FLAGS
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-DZHIP_WPH_MARK",
                           "-Wno-unused-function", "-Wno-unused-result", "-I" + CSRC, "-I" + os.path.join(ROOT, "include")] + UNIT_FLAGS["zhip_k_parse"]
                          + ["-D" + d for d in defs] + [srcp, "-o", asm], stderr=subprocess.DEVNULL)
    return asm


def kernel_body(asm):
    lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN4zhip5k_one\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    desc = {}
    for l in lines[start:end + 40]:
        m = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|private_segment_fixed_size)\s+(\S+)", l)
        if m: desc[m.group(1)] = m.group(2)
    for l in lines:
        m = re.match(r";\s*(NumVgprs|NumSgprs|ScratchSize|Occupancy):\s*(\d+)", l.strip())
        if m and m.group(1) not in desc: desc[m.group(1)] = m.group(2)
    return lines[start + 1:end], desc


def blocks_of(body):
    """[(label, [items])], items = ('ins', text) | ('mark', id)"""
    blocks, cur, label = [], [], "<entry>"
    for raw in body:
        m = re.match(r"^\s*;\s*ZWPH\s+(\d+)", raw)
        if m: cur.append(("mark", int(m.group(1)))); continue
        l = raw.split(";")[0].rstrip()
        if not l.strip(): continue
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: blocks.append((label, cur)); label, cur = m.group(1), []; continue
        if l.lstrip().startswith("."): continue
        cur.append(("ins", l.strip()))
    blocks.append((label, cur))
    return blocks


def table(mls=6, defs=()):
    body, desc = kernel_body(compile_asm(mls, defs))
    blocks = blocks_of(body)
    idx = {lab: i for i, (lab, _) in enumerate(blocks)}
    succ = []
    for i, (lab, items) in enumerate(blocks):
        s, fall = [], True
        for kind, t in items:
            if kind != "ins": continue
            op = t.split()[0]
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = t.split()[-1]
                if tgt in idx: s.append(idx[tgt])
                if op == "s_branch": fall = False
            if op in ("s_endpgm", "s_setpc_b64"): fall = False
        if fall and i + 1 < len(blocks): s.append(i + 1)
        succ.append(s)
    # backward pass: the phase(s) a block's tail belongs to = the first marker reachable from its end
    first = [next((t for k, t in items if k == "mark"), None) for _, items in blocks]
    tail = [frozenset() for _ in blocks]
    changed = True
    while changed:
        changed = False
        for i in range(len(blocks) - 1, -1, -1):
            new = set()
            for j in succ[i]:
                new |= ({first[j]} if first[j] is not None else tail[j])
            new = frozenset(new)
            if new != tail[i]: tail[i] = new; changed = True
    per = collections.defaultdict(collections.Counter)
    global LISTING
    LISTING = []
    for i, (lab, items) in enumerate(blocks):
        rows = []
        # walk backwards through the block: everything after the last marker belongs to tail[i], before a marker to that marker
        curph = tail[i]
        for kind, t in reversed(items):
            if kind == "mark": curph = frozenset({t}); continue
            key = "|".join(str(x) for x in sorted(curph)) if curph else "none"
            per[key][classify(t)] += 1
            rows.append((key, t))
        LISTING.append((lab, succ[i], rows[::-1]))
    return per, desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mls", type=int, default=6)
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--hits", help="JSON of scripts/prof_phases.py (window_phases: id -> [ticks per unit, visits per unit])")
    ap.add_argument("-D", action="append", default=[])
    ap.add_argument("--dump", help="print the instructions attributed to this phase key (e.g. 5 or 5|6), block by block")
    a = ap.parse_args()
    per, desc = table(a.mls, a.D)
    if a.dump:
        for bi, (lab, su, rows) in enumerate(LISTING):
            sel = [t for k, t in rows if k == a.dump]
            if sel:
                print("== block %d %s -> %s" % (bi, lab, su))
                for t in sel: print("    " + t)
        return
    if a.json:
        print(json.dumps({"desc": desc, "phases": {k: dict(v) for k, v in per.items()}})); return
    hits = None
    if a.hits:
        hits = json.load(open(a.hits))
    print("ZSTD_fast window, minMatch %d, static instructions per phase (one visit of a straight-line phase executes about this many); kernel: %s" % (a.mls, desc))
    hdr = "%-62s" % "phase" + " ".join("%6s" % c for c in COLS) + " |  total"
    if hits: hdr += " | visits/unit  ticks/unit ticks/visit | wave-instr/unit  per source byte"
    print(hdr)
    def key(k): return (0, int(k)) if k.isdigit() else (1, 0)
    tot = collections.Counter(); dynTot = 0.0
    for k in sorted(per, key=lambda k: (key(k), k)):
        c = per[k]; tot.update(c)
        name = ("%2s " % k + NAMES[int(k)]) if k.isdigit() else ("   shared by phases " + k)
        row = "%-62s" % name[:62] + " ".join("%6d" % c[x] for x in COLS) + " | %6d" % sum(c.values())
        if hits and k.isdigit() and k in hits["phases"]:
            t, v = hits["phases"][k]
            n = sum(c.values()) - c["wait"] - c["nop"]
            dyn = n * v; dynTot += dyn
            row += " | %11.1f %11.0f %11.0f | %15.0f %15.3f" % (v, t, t / v if v else 0, dyn, dyn / hits["unit_bytes"])
        print(row)
    print("%-62s" % "kernel total" + " ".join("%6d" % tot[x] for x in COLS) + " | %6d" % sum(tot.values()))
    if hits: print("sum over the phases with visits: %.0f wave-instructions per unit = %.3f per source byte (far calls, group loops and late inserts count once per visit)" % (dynTot, dynTot / hits["unit_bytes"]))


if __name__ == "__main__":
    main()
