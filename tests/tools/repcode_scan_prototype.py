#!/usr/bin/env python3
"""tests/tools/repcode_scan_prototype.py — design check for the block-parallel decoder, CPU only.

A block's sequences can be decoded without knowing the repcode history it starts from if offsets that name a repcode stay SYMBOLIC:
every history slot is either a constant (an offset introduced inside the block) or "incoming slot i minus d" (d > 0 only through the
`rep1 - 1` code).  A block's effect on the three-slot history is then a map of three such terms, maps compose associatively, an
exclusive scan over the blocks gives each block its incoming history, and one substitution pass makes the offsets concrete.
This script cuts the oracle's sequences of real inputs into pseudo-blocks, runs that scheme and compares with the sequential rule
(RFC 8878 3.1.1.5 / lib/decompress/zstd_decompress_block.c:1228-1290)."""
import ctypes as C
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _libs import load_oracle, datagen, text_like, _buf

CONST, IN = 0, 1


def term_eval(t, hist):
    return t[1] if t[0] == CONST else hist[t[1]] - t[2]


def step(slots, off_base, ll):
    """one sequence on a symbolic history `slots` (3 terms); returns (offset term, new slots) — the decoder's rule"""
    if off_base > 3:
        t = (CONST, off_base - 3, 0)
        return t, [t, slots[0], slots[1]]
    idx = off_base - 1 + (1 if ll == 0 else 0)              # 0..3
    if idx == 0:
        return slots[0], slots
    if idx == 3:
        b = slots[0]
        t = (CONST, b[1] - 1, 0) if b[0] == CONST else (IN, b[1], b[2] + 1)
        return t, [t, slots[0], slots[1]]
    t = slots[idx]
    return t, ([t, slots[0], slots[2]] if idx == 1 else [t, slots[0], slots[1]])


def compose(first, second):
    """history map of `first` followed by `second`"""
    out = []
    for t in second:
        if t[0] == CONST:
            out.append(t)
        else:
            b = first[t[1]]
            out.append((CONST, b[1] - t[2], 0) if b[0] == CONST else (IN, b[1], b[2] + t[2]))
    return out


def check(seqs, block_len, hist0=(1, 4, 8)):
    # sequential truth
    hist = list(hist0); truth = []
    for off_base, ll in seqs:
        t, hist_terms = step([(CONST, h, 0) for h in hist], off_base, ll)
        truth.append(t[1]); hist = [x[1] for x in hist_terms]
    # block-parallel: symbolic decode per block, scan of the maps, substitution
    blocks = [seqs[i:i + block_len] for i in range(0, len(seqs), block_len)]
    sym, maps = [], []
    for b in blocks:                                         # independent of each other
        slots = [(IN, 0, 0), (IN, 1, 0), (IN, 2, 0)]; terms = []
        for off_base, ll in b:
            t, slots = step(slots, off_base, ll); terms.append(t)
        sym.append(terms); maps.append(slots)
    incoming, acc = [], [(IN, 0, 0), (IN, 1, 0), (IN, 2, 0)]
    for m in maps:                                           # the scan (serial here; the operator is associative)
        incoming.append([term_eval(t, hist0) for t in acc]); acc = compose(acc, m)
    got = [term_eval(t, inc) for terms, inc in zip(sym, incoming) for t in terms]
    # associativity spot check: ((a.b).c) == (a.(b.c)) on the first maps
    if len(maps) >= 3:
        assert compose(compose(maps[0], maps[1]), maps[2]) == compose(maps[0], compose(maps[1], maps[2]))
    symbolic = sum(1 for terms in sym for t in terms if t[0] == IN)
    return got == truth, symbolic, len(seqs)


def main():
    lo = load_oracle()
    lo.zo_sequences_public.restype = C.c_size_t
    lo.zo_sequences_public.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    ok_all = True
    for name, a in (("datagen P50", datagen(lo, 131072, 50, 1)), ("datagen P90", datagen(lo, 131072, 90, 2)), ("text", text_like(131072, 3))):
        for level in (1, 3, 5, 7):
            cp = (C.c_uint * 7)()
            lo.zo_get_cparams(level, len(a), cp)
            out = np.zeros((len(a) // 3 + 8) * 4, dtype=np.uint32)
            n = lo.zo_sequences_public(cp, _buf(a), len(a), _buf(out), len(a) // 3 + 8)
            S = out[:4 * n].reshape(-1, 4)
            seqs = [(int(r) if r else int(o) + 3, int(ll)) for o, ll, ml, r in S]
            for bl in (1, 7, 64, 1000):
                ok, symbolic, total = check(seqs, bl)
                ok_all = ok_all and ok
                print(f"{name:12s} L{level} blocks of {bl:4d} sequences: {'ok' if ok else 'MISMATCH'}  ({symbolic} of {total} offsets symbolic before the scan)")
    print("all ok" if ok_all else "FAILED")
    return 0 if ok_all else 1


if __name__ == "__main__":
    sys.exit(main())
