"""Structured-random parity fuzz: the product kernels on the host SIMT emulator vs the oracle, whole frames, many small
inputs across levels (the reference's own tests/fuzz/ idea applied to the device path).  Seeds are fixed: failures reproduce."""
import os
import numpy as np
import pytest
from _libs import load_oracle, load_emu, emu_compress_units, _buf, ERR

LEVELS = (-5, -1, 1, 2, 3, 4, 5, 6, 7, 9)


def gen(rng, n):
    """mix of literal noise, back-references at random distances, runs and small alphabets"""
    out = np.zeros(n, dtype=np.uint8)
    pos = 0
    alpha = int(rng.choice([2, 4, 16, 64, 256]))
    while pos < n:
        kind = rng.integers(0, 10)
        ln = int(min(n - pos, rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 31, 64, 130, 400, 1500])))
        if kind < 4 or pos < 8:
            out[pos:pos + ln] = rng.integers(0, alpha, size=ln, dtype=np.uint8)
        elif kind < 8:
            off = int(rng.integers(1, pos + 1)) if rng.random() < 0.7 else int(rng.integers(1, min(pos, 16) + 1))
            for i in range(ln):                      # overlapping copy, like a real match
                out[pos + i] = out[pos + i - off]
        else:
            out[pos:pos + ln] = rng.integers(0, 256)
        pos += ln
    return out


def oracle_unit(lo, a, level):
    cap = lo.zo_compress_bound(len(a)) + 64
    dst = np.zeros(cap, dtype=np.uint8)
    r = lo.zo_compress_unit(_buf(dst), cap, _buf(a), len(a), level)
    return None if r == ERR else dst[:r].tobytes()


def run(seed0, rounds, per_round, maxlen):
    lo, le = load_oracle(), load_emu()
    bad = []
    for rd in range(rounds):
        rng = np.random.default_rng(seed0 + rd)
        level = LEVELS[(seed0 + rd) % len(LEVELS)]
        bufs = [gen(rng, int(rng.integers(0, maxlen))) for _ in range(per_round)]
        import ctypes as C
        def strat(n):
            cp = (C.c_uint * 7)()
            return cp[6] if lo.zo_get_cparams(level, n, cp) == 0 else -1
        groups = {}
        for b in bufs:                               # the emulator harness wants one strategy family per launch
            s = strat(len(b))
            if 1 <= s <= 5:
                groups.setdefault(min(s, 3), []).append(b)
        for fam, bs in groups.items():
            frames = emu_compress_units(le, lo, bs, level)
            for b, f in zip(bs, frames):
                if f != oracle_unit(lo, b, level):
                    bad.append((seed0 + rd, level, len(b)))
    return bad


def test_fuzz_small_units_all_levels():
    bad = run(1000, 40, 24, 2600)
    assert not bad, bad[:10]


@pytest.mark.skipif(not os.environ.get("ZHIP_LONG_FUZZ"), reason="set ZHIP_LONG_FUZZ=1 for the long run")
def test_fuzz_long():
    bad = run(50000, int(os.environ.get("ZHIP_LONG_FUZZ_ROUNDS", "600")), 32, 9000)
    assert not bad, bad[:10]


def run_dict(seed0, rounds, per_round):
    import ctypes as C
    from _libs import UNIT_DT, SEQ_DT, PARSE_DT
    lo, le = load_oracle(), load_emu()
    lo.zo_cdict_create.restype = C.c_void_p
    lo.zo_cdict_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    lo.zo_cdict_free.argtypes = [C.c_void_p]
    lo.zo_parse_cdict.restype = C.c_size_t
    lo.zo_parse_cdict.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    le.emu_parse_dict.restype = C.c_int
    le.emu_parse_dict.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t, C.c_int] + [C.c_void_p] * 4 + [C.c_int]
    bad = []
    for rd in range(rounds):
        rng = np.random.default_rng(seed0 + rd)
        level = (-3, 1, 2, 3, 4)[(seed0 + rd) % 5]
        dict_ = gen(rng, int(rng.integers(8, 70000)))
        cutoff = 8192 if level < 3 else 16384
        recs = []
        for k in range(per_round):
            n = int(rng.integers(8, min(cutoff, 5000)))
            if k % 6 == 5:
                n = int(rng.integers(cutoff + 1, cutoff + 30000))      # above the attach cut-off: the dictionary's COPY mode (k_parse_ext)
            r = gen(rng, n)
            for _ in range(int(rng.integers(0, 6))):          # splice in pieces of the dictionary (incl. its very end)
                ln = int(rng.integers(4, 200)); ln = min(ln, n, len(dict_))
                s = len(dict_) - ln if rng.random() < 0.2 else int(rng.integers(0, len(dict_) - ln + 1))
                d = int(rng.integers(0, n - ln + 1))
                r[d:d + ln] = dict_[s:s + ln]
            recs.append(r)
        cd = lo.zo_cdict_create(_buf(dict_), len(dict_), level)
        if not cd:
            continue
        offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.uint64)
        src = np.concatenate(recs + [np.zeros(16, np.uint8)])
        nrec = len(recs); cap = le.emu_seq_cap(); lstride = le.emu_lit_stride()
        units = np.zeros(nrec, dtype=UNIT_DT); seqs = np.zeros(nrec * cap, dtype=SEQ_DT); metas = np.zeros(nrec, dtype=PARSE_DT)
        lits = np.full(nrec * lstride, 0xEE, dtype=np.uint8)
        rc = le.emu_parse_dict(_buf(src), _buf(offs), nrec, _buf(dict_), len(dict_), level, _buf(units), _buf(seqs), _buf(lits), _buf(metas), 0)
        if rc != 0:
            lo.zo_cdict_free(cd); continue
        for i, r in enumerate(recs):
            want = np.zeros((len(r) // 3 + 8, 3), dtype=np.uint32)
            nw = lo.zo_parse_cdict(cd, _buf(r), len(r), _buf(want), len(want))
            m = metas[i]; s = seqs[i * cap: i * cap + int(m["nbSeq"])]
            got = np.stack([s["litLength"].astype(np.uint32), s["mlBase"].astype(np.uint32) + 3, s["offBase"]], axis=1) if len(s) else np.zeros((0, 3), np.uint32)
            if len(got) != nw or (got != want[:nw]).any():
                bad.append((seed0 + rd, level, i, len(r), len(dict_)))
        lo.zo_cdict_free(cd)
    return bad


def test_fuzz_dictionary_records():
    bad = run_dict(7000, 25, 12)
    assert not bad, bad[:10]


def test_fuzz_decoder_on_the_emulator():
    """the device decoder on frames of the oracle's compressor (all implemented levels): structured-random inputs, ragged sizes"""
    import ctypes as C
    from test_emu_decode import emu_decode, DFRAME_DT
    lo, le = load_oracle(), load_emu()
    le.emu_decode.restype = C.c_int
    le.emu_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint, C.c_int]
    for rd in range(30):
        rng = np.random.default_rng(31000 + rd)
        level = LEVELS[rd % len(LEVELS)]
        bufs = [gen(rng, int(rng.integers(0, 9000))) for _ in range(20)] + ([gen(rng, int(rng.integers(60000, 131073)))] if rd % 5 == 0 else [])
        frames, want = [], []
        for b in bufs:
            f = oracle_unit(lo, b, level)
            if f is not None:
                frames.append(f); want.append(b.tobytes())
        got = emu_decode(le, frames, [len(w) for w in want], groups=5)
        for i, ((st, data, _), w) in enumerate(zip(got, want)):
            assert st == 0 and data == w, (rd, level, i, len(w), st)


def _explicit(level, n, req):
    """the product's host-side parameter logic (no GPU needed): requested -> effective parameters, None when not for the device"""
    import ctypes as C
    import zstd_amd
    L = zstd_amd.lib()
    L.zhip_getCParams_explicit.restype = C.c_int
    L.zhip_getCParams_explicit.argtypes = [C.c_int, C.c_ulonglong, C.c_void_p, C.c_void_p]
    eff = (C.c_uint * 7)()
    if L.zhip_getCParams_explicit(level, n, (C.c_uint * 7)(*req), eff) != 0:
        return None
    return eff


def test_fuzz_units_with_explicit_parameters():
    """random explicit parameters (table logs 8-17, searchLog 1-6, minMatch 3-7, targetLength, every strategy up to lazy2, row matcher
    on / off) on structured-random units: the kernels on the emulator against the oracle (the same sweep against the real reference:
    500 trials clean when this test was written, /root/reference is not needed here)"""
    import ctypes as C
    import _libs
    lo, le = load_oracle(), load_emu()
    lo.zo_compress_unit_params.restype = C.c_size_t
    lo.zo_compress_unit_params.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    rng = np.random.default_rng(77)
    orig = _libs.make_units
    seen = 0
    try:
        for t in range(60):
            n = int(rng.integers(1, 30000)) if t % 6 else int(rng.integers(100000, 131073))
            a = gen(rng, n)
            level = int(rng.choice([1, 3, 5, 6, 7, -3]))
            req = [int(rng.choice([0, 0, 12, 14, 15, 17, 18])), int(rng.choice([0, 0, 8, 12, 15, 16])), int(rng.choice([0, 0, 8, 11, 13, 15, 17, 18])),
                   int(rng.choice([0, 0, 1, 2, 4, 5, 6])), int(rng.choice([0, 0, 3, 4, 5, 6, 7])), int(rng.choice([0, 0, 1, 4, 16, 64])), int(rng.choice([0, 0, 1, 2, 3, 4, 5]))]
            eff = _explicit(level, n, req)
            if eff is None or not (1 <= eff[6] <= 5) or (1 << eff[0]) < n or (eff[6] == 1 and eff[2] > 15):
                continue
            no_row = int(rng.integers(0, 2))
            row = 3 <= eff[6] <= 5 and eff[0] > 14 and not no_row
            lo.zo_set_row_matcher(1 if row else 0)
            cap = lo.zo_compress_bound(n) + 64
            o = np.zeros(cap, dtype=np.uint8)
            r = lo.zo_compress_unit_params(_buf(o), cap, _buf(a), n, eff)
            assert r != ERR

            def mk(lo_, sizes, level_, unit=131072, row=False, eff=eff, no_row=no_row):
                units = orig(lo_, sizes, 1, unit, False)
                for f in units:
                    f["windowLog"], f["chainLog"], f["hashLog"], f["searchLog"], f["minMatch"], f["targetLength"], f["strategy"] = list(eff)
                    f["litMode"] = 1 if (eff[6] == 1 and eff[5] > 0) else 0
                    f["rowLog"] = min(6, max(4, eff[3])) if (3 <= eff[6] <= 5 and eff[0] > 14 and not no_row) else 0
                return units
            _libs.make_units = mk
            got = emu_compress_units(le, lo, [a], 1)[0]
            _libs.make_units = orig
            assert got == o[:r].tobytes(), (t, n, level, req, list(eff), no_row)
            seen += 1
    finally:
        _libs.make_units = orig
        lo.zo_set_row_matcher(0)
    assert seen >= 30


def test_lazy_units_at_hashlog_18_on_the_emulator():
    """hashLog 18 (windowLog + 1 of a 128 KB unit) splits the hash-chain builder's table into SIXTEEN LDS slices; until round 6 its slice counters held eight and the
    case was never run (the fuzzers stopped at 17).  Hash chain and row matcher, greedy / lazy / lazy2, against the oracle."""
    import ctypes as C
    import _libs
    lo, le = load_oracle(), load_emu()
    lo.zo_compress_unit_params.restype = C.c_size_t
    lo.zo_compress_unit_params.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    rng = np.random.default_rng(1818)
    orig = _libs.make_units
    ran = 0
    try:
        for strat, no_row, n in ((3, 1, 131072), (4, 0, 131072), (5, 1, 100003), (3, 0, 70000)):
            a = gen(rng, n)
            req = [18, 16, 18, 4, 5, 16, strat]
            eff = _explicit(5, n, req)
            assert eff is not None and eff[2] == 18, list(eff)
            row = eff[0] > 14 and not no_row
            lo.zo_set_row_matcher(1 if row else 0)
            cap = lo.zo_compress_bound(n) + 64
            o = np.zeros(cap, dtype=np.uint8)
            r = lo.zo_compress_unit_params(_buf(o), cap, _buf(a), n, eff)
            assert r != ERR

            def mk(lo_, sizes, level_, unit=131072, row=False, eff=eff, no_row=no_row):
                units = orig(lo_, sizes, 1, unit, False)
                for f in units:
                    f["windowLog"], f["chainLog"], f["hashLog"], f["searchLog"], f["minMatch"], f["targetLength"], f["strategy"] = list(eff)
                    f["litMode"] = 0
                    f["rowLog"] = min(6, max(4, eff[3])) if (eff[0] > 14 and not no_row) else 0
                return units
            _libs.make_units = mk
            got = emu_compress_units(le, lo, [a], 1)[0]
            _libs.make_units = orig
            assert got == o[:r].tobytes(), (strat, no_row, n, list(eff))
            ran += 1
    finally:
        _libs.make_units = orig
        lo.zo_set_row_matcher(0)
    assert ran == 4
