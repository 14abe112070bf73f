"""Oracle vs committed golden vectors (made by tests/golden/make_golden.py with the real reference).
Runs anywhere (no /root/reference, no oracle/_ref needed)."""
import hashlib, json, os
import numpy as np
from _libs import load_oracle, corpus_cases, _buf, ERR

GOLD = os.path.join(os.path.dirname(__file__), "golden", "units_v1.json")


GOLD_HC = os.path.join(os.path.dirname(__file__), "golden", "units_v2_hashchain.json")


def test_oracle_reproduces_golden_units():
    _check(GOLD, (1, 3), 400)


def test_oracle_reproduces_golden_hashchain_units():
    """greedy / lazy / lazy2 with the hash-chain matcher (the reference run with useRowMatchFinder disabled)"""
    _check(GOLD_HC, (5, 6, 7), 600)


def _check(path, levels, atleast):
    lo = load_oracle()
    gold = {(g["case"], g["level"]): g for g in json.load(open(path))["units"]}
    sizes = sorted({g["n"] for g in gold.values()})
    seen = 0
    for n in sizes:
        for name, a in corpus_cases(lo, sizes=(n,), seeds=(0, 5)):
            for level in levels:
                g = gold.get((name, level))
                if g is None:
                    continue
                assert hashlib.sha256(a.tobytes()).hexdigest() == g["src_sha256"], name
                cap = lo.zo_compress_bound(n) + 64
                dst = np.zeros(cap, dtype=np.uint8)
                r = lo.zo_compress_unit(_buf(dst), cap, _buf(a), n, level)
                assert r != ERR and r == g["csize"], (name, level)
                assert hashlib.sha256(dst[:r].tobytes()).hexdigest() == g["dst_sha256"], (name, level)
                seen += 1
    assert seen >= len(gold) > atleast
