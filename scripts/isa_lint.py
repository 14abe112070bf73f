#!/usr/bin/env python3
"""scripts/isa_lint.py [--json] [unit ...] — static checks of the compiled gfx950 code, one translation unit at a time (zhip_k_*.hip -> assembly).

Rule B (barriers): no s_barrier may execute while EXEC is narrowed.  A workgroup barrier under a lane mask means the compiler treats
the control flow around it as lane-divergent, and then the ORDER in which it lays out the "divergent" paths decides whether the kernel
works: round 2's table prefix fill and round 3's decoder both stalled on the GPU that way while passing on the host emulator (DESIGN.md
4.7b / 4.6c).  The check is a forward data-flow pass over the kernel's control-flow graph: the state is the set of SGPR pairs that hold a
saved EXEC (s_and_saveexec & co. add, `s_or_b64 exec, exec, <pair>` removes); a barrier reached with a non-empty set is reported.
The fix is always the same: make the branch condition scalar (ZHIP_UNIFORM) and separate leader-only regions with ZHIP_CONVERGE.

Rule C (calls): k_decode makes no function call (its helpers are always_inline; an outlined Huffman stage was round 3's first suspect).

Also prints per kernel: VGPRs, SGPRs, scratch bytes, LDS bytes, barriers, and a hash of the instruction stream (labels and comments
stripped) — tests/test_isa_checks.py compares the hashes with tests/golden/isa_pins.json, the code that last passed on a real MI355X."""
import hashlib, json, os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zstd_amd", "csrc")
UNITS = ["zhip_k_parse", "zhip_k_lazy", "zhip_k_entropy", "zhip_k_frames", "zhip_k_decode"]
PAIR = r"(s\[\d+:\d+\]|vcc|-1|0)"


def compile_asm(unit):
    out = os.path.join(tempfile.gettempdir(), f"isa_lint_{unit}.s")
    sys.path.insert(0, ROOT)
    from zstd_amd.build import UNIT_FLAGS                         # the unit's own code generation options: the lint reads the code the library ships
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                           "-Wno-unused-function", "-Wno-unused-result"] + UNIT_FLAGS.get(unit, []) + [os.path.join(CSRC, unit + ".hip"), "-o", out], stderr=subprocess.DEVNULL)
    return out


def kernels_of(path):
    """yield (name, lines of the function body, descriptor dict)"""
    text = open(path).read().split("\n")
    i = 0
    while i < len(text):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", text[i])
        if m and not text[i].startswith(".L"):
            name = m.group(1)
            j = i + 1
            while j < len(text) and not text[j].startswith(".Lfunc_end"):
                j += 1
            body = text[i + 1:j]
            desc = {}
            for l in body:                              # the kernel descriptor sits between s_endpgm and .Lfunc_end
                mm = re.match(r"\s*\.amdhsa_(\w+)\s+(\S+)", l)
                if mm:
                    desc[mm.group(1)] = mm.group(2)
            if desc:                                   # a kernel (device functions have no descriptor)
                yield name, body, desc
            i = j
        i += 1


def short(name):
    m = re.match(r"_ZN4zhip(\d+)", name)
    return name[8 + len(m.group(1)):8 + len(m.group(1)) + int(m.group(1))] if m else name


def blocks_of(body):
    """basic blocks: list of (label or None, [instructions]); successors by label / fall-through"""
    blocks, cur, label = [], [], "<entry>"
    for l in body:
        l = l.split(";")[0].rstrip()
        if not l.strip():
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((label, cur)); label, cur = m.group(1), []
            continue
        if l.startswith("\t.") or l.startswith("."):
            continue
        cur.append(l.strip())
    blocks.append((label, cur))
    return blocks


def barrier_rule(body):
    blocks = blocks_of(body)
    index = {lab: i for i, (lab, _) in enumerate(blocks)}
    succ = []
    for i, (lab, ins) in enumerate(blocks):
        s, fall = [], True
        for x in ins:
            m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", x)
            if m:
                s.append(index[m.group(1)])
            m = re.match(r"s_branch\s+(\.LBB\d+_\d+)", x)
            if m:
                s.append(index[m.group(1)]); fall = False
            if x.startswith("s_endpgm"):
                fall = False
        if fall and i + 1 < len(blocks):
            s.append(i + 1)
        succ.append(s)

    def transfer(state, ins, report=None, lab=None):
        st = list(state)                                # a stack of saved-EXEC pairs, outermost first
        for x in ins:
            if x.startswith("s_barrier") and st and report is not None:
                report.append((lab, list(st)))
            m = re.match(r"s_(?:and|andn2|or|orn2|xor|nand|nor|xnor|andn1|orn1)_saveexec_b64\s+" + PAIR, x)
            if m:
                if m.group(1) in st: del st[st.index(m.group(1)):]           # the else-half of an if/else re-saves into the same pair
                st.append(m.group(1)); continue
            m = re.match(r"s_or_b64\s+exec,\s*exec,\s*" + PAIR, x) or re.match(r"s_mov_b64\s+exec,\s*" + PAIR, x)
            if m:
                q = m.group(1)
                if q == "-1": st = []
                elif q in st: del st[st.index(q):]       # back to what that pair saved: it and everything narrowed inside it are over
                elif x.startswith("s_or") and st: st.pop()                  # a mask kept under another name (copied / reloaded from a spill lane)
                elif x.startswith("s_mov"): st.append("anon")               # exec = a computed mask (the compiler's own if-conversion)
                continue
            if re.match(r"s_(?:and|andn2|xor)_b64\s+exec,", x):
                if not st: st.append("anon")            # narrowing without a save of its own (loop masks): over at the next restore
                continue
            m = re.match(r"[sv]_\w+\s+(s\[(\d+):(\d+)\]|s(\d+)|vcc)\b", x)      # an overwritten pair no longer holds a saved EXEC (pairs are reused all the time)
            if m and st:
                if m.group(1) == "vcc": lo = hi = -1
                elif m.group(2): lo, hi = int(m.group(2)), int(m.group(3))
                else: lo = hi = int(m.group(4))
                def hit(q):
                    if q == "vcc": return m.group(1) == "vcc"
                    if not q.startswith("s["): return False
                    a_, b_ = map(int, q[2:-1].split(":"))
                    return not (hi < a_ or lo > b_)
                st = [q for q in st if not hit(q)]
        return tuple(st)

    def meet(a, b):                                     # what is narrowed on EVERY path into a block: the common prefix
        n = 0
        while n < len(a) and n < len(b) and a[n] == b[n]:
            n += 1
        return a[:n]

    def barrier_before_restore(ins):                    # the block runs into s_barrier before it touches EXEC
        for x in ins:
            if x.startswith("s_barrier"): return True
            if re.search(r"\bexec\b", x.split(",")[0]): return False
        return False

    inn = [None] * len(blocks); inn[0] = ()
    mismatch = {}
    work = [0]
    while work:
        b = work.pop()
        out = transfer(inn[b], blocks[b][1])
        for t in succ[b]:
            if inn[t] is not None and out != inn[t] and barrier_before_restore(blocks[t][1]):
                mismatch[blocks[t][0]] = ["paths disagree: %s / %s" % ("+".join(out) or "full", "+".join(inn[t]) or "full")]
            new = out if inn[t] is None else meet(inn[t], out)
            if inn[t] is None or new != inn[t]:
                inn[t] = new; work.append(t)
    report = []
    for b, (lab, ins) in enumerate(blocks):
        if inn[b] is not None:
            transfer(inn[b], ins, report, lab)
    return report, sorted(mismatch.items())


def stream_hash(body):
    h = hashlib.sha256()
    for l in body:
        l = l.split(";")[0].strip()
        if not l or l.startswith("."):
            continue
        h.update(re.sub(r"\.LBB\d+_\d+", "L", l).encode() + b"\n")
    return h.hexdigest()[:16]


def lint(units=UNITS):
    with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 2)) as ex:
        paths = list(ex.map(compile_asm, units))
    res = {}
    for unit, path in zip(units, paths):
        for name, body, desc in kernels_of(path):
            k = short(name)
            res[k] = {"unit": unit, "vgpr": int(desc.get("next_free_vgpr", 0)), "sgpr": int(desc.get("next_free_sgpr", 0)),
                      "scratch": int(desc.get("private_segment_fixed_size", 0)), "lds": int(desc.get("group_segment_fixed_size", 0)),
                      "barriers": sum(1 for l in body if re.match(r"\s*s_barrier", l)),
                      "calls": sum(1 for l in body if "s_swappc_b64" in l),
                      "hash": stream_hash(body)}
            must, notes = barrier_rule(body)
            res[k]["masked_barriers"] = [f"{lab}: {'+'.join(st)}" for lab, st in must]     # narrowed on EVERY path into the barrier
            res[k]["path_notes"] = [f"{lab}: {st[0]}" for lab, st in notes]                 # paths arrive with different masks (the tracking is coarse: informational)
    return res


PINS = os.path.join(ROOT, "tests", "golden", "isa_pins.json")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    res = lint(args or UNITS)
    if "--pin" in sys.argv:                             # after a green `pytest -m gpu` on a real MI355X: these instruction streams are the tested ones
        pins = {k: r["hash"] for k, r in sorted(res.items()) if r["barriers"]}
        json.dump({"note": "instruction-stream hashes of the kernels that use workgroup barriers, as they last passed pytest -m gpu on MI355X "
                           "(scripts/isa_lint.py --pin; ROCm 7.2.0 hipcc -O3)", "kernels": pins}, open(PINS, "w"), indent=1)
        print("pinned", len(pins), "kernels ->", PINS)
    if "--json" in sys.argv:
        print(json.dumps(res, indent=1, sort_keys=True))
    else:
        bad = 0
        for k in sorted(res):
            r = res[k]
            print(f"{k:20s} {r['unit'][7:]:8s} vgpr {r['vgpr']:3d} sgpr {r['sgpr']:3d} scratch {r['scratch']:4d} barriers {r['barriers']:2d} calls {r['calls']} hash {r['hash']}"
                  + (f"  MASKED BARRIERS: {len(r['masked_barriers'])}" if r["masked_barriers"] else ""))
            for mb in r["masked_barriers"][:6]:
                print("       masked:", mb)
            for mb in r["path_notes"][:4]:
                print("       note:  ", mb)
            bad += len(r["masked_barriers"])
        print("s_barrier under a narrowed EXEC on every path:", bad)
        print("k_decode function calls:", res.get("k_decode", {}).get("calls", "n/a"))
