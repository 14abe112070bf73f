"""Product parse kernel (zstd_amd/csrc/zhip_parse.h) executed on the host SIMT emulator vs the oracle's sequences.
CPU only: this checks the wave-parallel algorithm, not the GPU build."""
import ctypes as C
import numpy as np
import pytest
from _libs import (load_oracle, load_emu, load_ref, have_ref, lorem, corpus_cases, make_units, oracle_parse, emu_parse_units, emu_compress_units, _buf, ERR, SEQ_DT, PARSE_DT, UNIT_DT as UNIT_DT_, datagen, text_like)


@pytest.fixture(scope="module")
def libs():
    return load_oracle(), load_emu()


def emu_parse(le, lo, bufs, level):
    sizes = [len(b) for b in bufs]
    units = make_units(lo, sizes, level)
    src = np.concatenate(bufs + [np.zeros(16, dtype=np.uint8)]) if bufs else np.zeros(16, dtype=np.uint8)
    cap = le.emu_seq_cap()
    seqs = np.zeros(len(bufs) * cap, dtype=SEQ_DT)
    metas = np.zeros(len(bufs), dtype=PARSE_DT)
    lstride = le.emu_lit_stride()
    lits = np.full(max(1, len(bufs)) * lstride, 0xEE, dtype=np.uint8)
    emu_parse_units(le, src, units, seqs, lits, metas)
    out = []
    for i in range(len(bufs)):
        m = metas[i]
        m_lits = lits[i * lstride: i * lstride + int(m["litSize"])]
        s = seqs[i * cap: i * cap + int(m["nbSeq"])]
        ll = s["litLength"].astype(np.uint32)
        ml = s["mlBase"].astype(np.uint32) + 3
        if m["longType"] == 1:
            ll[m["longPos"]] += 0x10000
        if m["longType"] == 2:
            ml[m["longPos"]] += 0x10000
        out.append((np.stack([ll, ml, s["offBase"]], axis=1) if len(s) else np.zeros((0, 3), np.uint32), m, m_lits))
    return out


def check(le, lo, cases, level):
    names = [c[0] for c in cases]
    bufs = [c[1] for c in cases]
    res = emu_parse(le, lo, bufs, level)
    for name, a, (seqs, m, glits) in zip(names, bufs, res):
        oseqs, litSize, rep, olits = oracle_parse(lo, a, level, want_lits=True)
        assert int(m["litSize"]) == litSize and np.array_equal(glits, olits), (name, "literal buffer differs")
        assert len(seqs) == len(oseqs), (name, len(seqs), len(oseqs))
        if len(seqs):
            bad = np.nonzero((seqs != oseqs).any(axis=1))[0]
            assert len(bad) == 0, (name, int(bad[0]), seqs[bad[0]].tolist(), oseqs[bad[0]].tolist())
        assert int(m["lastLits"]) == litSize - int(oseqs[:, 0].sum()), name
        assert list(m["rep"][:2]) == rep[:2], name


def test_fast_parse_small_sizes(libs):
    lo, le = libs
    cases = []
    for n in (0, 1, 7, 8, 12, 13, 14, 20, 33, 64, 100, 129, 200, 257, 1000, 4097):
        cases += list(corpus_cases(lo, sizes=(n,), seeds=(0,)))
    check(le, lo, cases, 1)


def test_fast_parse_128k(libs):
    lo, le = libs
    check(le, lo, list(corpus_cases(lo, sizes=(131072,), seeds=(0,))), 1)


def test_fast_parse_ragged_and_negative_levels(libs):
    lo, le = libs
    for level in (-1, -5, 2):
        cases = []
        for n in (5000, 16384, 40000, 100001):
            cases += list(corpus_cases(lo, sizes=(n,), seeds=(3,)))
        cases = [c for c in cases if make_units(lo, [len(c[1])], level)["strategy"][0] == 1]
        check(le, lo, cases, level)


def test_dfast_parse_levels_3_4(libs):
    lo, le = libs
    for level in (3, 4):
        cases = []
        for n in (0, 9, 10, 11, 20, 100, 1000, 5000, 40000, 131072):
            cases += list(corpus_cases(lo, sizes=(n,), seeds=(level,)))
        def strat(nn):
            cp = (C.c_uint * 7)()
            return cp[6] if lo.zo_get_cparams(level, nn, cp) == 0 else -1       # level 4 below 16 KB is greedy: not ours
        cases = [c for c in cases if strat(len(c[1])) == 2]
        assert cases
        check(le, lo, cases, level)


def _strat(lo, level, nn):
    cp = (C.c_uint * 7)()
    return cp[6] if lo.zo_get_cparams(level, nn, cp) == 0 else -1


@pytest.mark.parametrize("level", [5, 6, 7, 9])
def test_hashchain_parse_small(libs, level):
    """greedy / lazy / lazy2 (hash chain) on the emulator vs the oracle: small and ragged units"""
    lo, le = libs
    cases = []
    for n in (0, 9, 10, 11, 12, 20, 100, 1000, 5000, 20000):
        cases += list(corpus_cases(lo, sizes=(n,), seeds=(level,)))
    cases = [c for c in cases if 3 <= _strat(lo, level, len(c[1])) <= 5]
    assert cases
    check(le, lo, cases, level)


@pytest.mark.parametrize("level", [5, 6, 8])
def test_hashchain_parse_128k(libs, level):
    lo, le = libs
    check(le, lo, list(corpus_cases(lo, sizes=(131072,), seeds=(1,))), level)


@pytest.mark.parametrize("level", [5, 7])
def test_hashchain_parse_skip_regions_then_matches(libs, level):
    """lazy-skipping stretches (positions never inserted) followed by compressible data that points back into them"""
    lo, le = libs
    from _libs import datagen
    rng = np.random.default_rng(level)
    r1 = rng.integers(0, 256, size=20000, dtype=np.uint8)
    mixed = np.concatenate([r1, datagen(lo, 30000, 60, level), r1[3000:15000], rng.integers(0, 256, size=9000, dtype=np.uint8),
                            r1[:9000], datagen(lo, 21072, 30, level + 1)])
    check(le, lo, [("mixed", mixed), ("mixed_short", mixed[:70001])], level)


def test_dict_records_parse_like_the_oracle(libs):
    """k_parse_dict (dfast with an attached dictionary) on the emulator vs the oracle's dictMatchState restatement"""
    _dict_records_check(libs)


@pytest.mark.parametrize("mode", [1, 2])
def test_dict_records_queue_forms_on_the_emulator(libs, monkeypatch, mode):
    """the records stage as a ticket queue: k_parse_dict_q (tables in LDS) and k_parse_dict_g (tables in global memory, ballot hash groups)"""
    monkeypatch.setenv("ZHIP_EMU_DICT_QUEUE", str(mode))
    _dict_records_check(libs, cases=(("text", 3), ("datagen", 3), ("text", 1), ("text", -3)))


def _dict_records_check(libs, cases=(("text", 3), ("datagen", 3), ("text", 4), ("text", 1), ("datagen", 1), ("text", 2), ("text", -3))):
    lo, le = libs
    from _libs import datagen, text_like
    lo.zo_cdict_create.restype = C.c_void_p
    lo.zo_cdict_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    lo.zo_cdict_free.argtypes = [C.c_void_p]
    lo.zo_parse_cdict.restype = C.c_size_t
    lo.zo_parse_cdict.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    le.emu_parse_dict.restype = C.c_int
    le.emu_parse_dict.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t, C.c_int] + [C.c_void_p] * 4 + [C.c_int]
    rng = np.random.default_rng(5)
    for kind, level in cases:
        corpus = text_like(200000, 3) if kind == "text" else datagen(lo, 200000, 60, 3)
        dict_ = corpus[:110000 if level in (3, 1) else 60000].copy()
        recs = []
        # sizes above the attach cut-off (16 KB dfast / 8 KB fast) take the reference's COPY mode: k_ext_init + k_parse_ext
        for n in ((8, 9, 10, 17, 100, 500, 1000, 1024, 1500, 4000, 8000, 16384, 12000, 300, 64, 16385, 20000, 50000, 131072) if level >= 3 else (8, 9, 10, 12, 17, 100, 500, 1000, 1024, 1500, 4000, 8000, 8192, 300, 64, 8193, 9000, 40000, 131072)):
            s = int(rng.integers(0, len(corpus) - n))
            r = corpus[s:s + n].copy()
            if n > 50:
                k = rng.integers(0, n, size=max(1, n // 40))
                r[k] = rng.integers(0, 256, size=len(k), dtype=np.uint8)
            recs.append(r)
        recs.append(dict_[-300:].copy())                 # ends exactly like the dictionary: matches that run off its end
        recs.append(np.concatenate([dict_[-40:], dict_[:200], dict_[-40:]]))
        recs.append(np.concatenate([dict_[-9000:], dict_[:9000], dict_[-40:]]))          # copy mode: 2-segment matches off the dictionary's end
        offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.uint64)
        src = np.concatenate(recs + [np.zeros(16, np.uint8)])
        nrec = len(recs)
        cap = le.emu_seq_cap(); lstride = le.emu_lit_stride()
        units = np.zeros(nrec, dtype=UNIT_DT_)
        seqs = np.zeros(nrec * cap, dtype=SEQ_DT); metas = np.zeros(nrec, dtype=PARSE_DT)
        lits = np.full(nrec * lstride, 0xEE, dtype=np.uint8)
        rc = le.emu_parse_dict(_buf(src), _buf(offs), nrec, _buf(dict_), len(dict_), level, _buf(units), _buf(seqs), _buf(lits), _buf(metas), 0)
        assert rc == 0, rc
        cd = lo.zo_cdict_create(_buf(dict_), len(dict_), level)
        for i, r in enumerate(recs):
            want = np.zeros((len(r) // 3 + 8, 3), dtype=np.uint32)
            nw = lo.zo_parse_cdict(cd, _buf(r), len(r), _buf(want), len(want))
            m = metas[i]
            s = seqs[i * cap: i * cap + int(m["nbSeq"])]
            got = np.stack([s["litLength"].astype(np.uint32), s["mlBase"].astype(np.uint32) + 3, s["offBase"]], axis=1) if len(s) else np.zeros((0, 3), np.uint32)
            assert len(got) == nw, (kind, level, len(r), len(got), nw)
            bad = np.nonzero((got != want[:nw]).any(axis=1))[0]
            assert len(bad) == 0, (kind, level, len(r), int(bad[0]), got[bad[0]].tolist(), want[bad[0]].tolist())
        lo.zo_cdict_free(cd)


def test_fast_parse_explicit_parameters(libs):
    """explicit cParams (B2: ZSTD_c_hashLog / minMatch / targetLength set on the CCtx) reach the kernel through ZhipUnit: the
    reference's default level-1 row applied to 128 KB units (hashLog 14, minMatch 7), a 15-bit table with a 4-byte hash, and an
    accelerated scan (targetLength 3 -> step 4: the schedule-shaped batches only)"""
    lo, le = libs
    for cpv in ([17, 13, 14, 1, 7, 0, 1], [17, 12, 15, 1, 4, 0, 1], [17, 12, 13, 1, 5, 3, 1]):
        cases = list(corpus_cases(lo, sizes=(131072, 40000), seeds=(5,)))[:8]
        bufs = [c[1] for c in cases]
        units = make_units(lo, [len(b) for b in bufs], 1)
        for k, f in enumerate(("windowLog", "chainLog", "hashLog", "searchLog", "minMatch", "targetLength", "strategy")):
            units[f] = cpv[k]
        units["litMode"] = 1 if cpv[5] > 0 else 0
        src = np.concatenate(bufs + [np.zeros(16, dtype=np.uint8)])
        cap = le.emu_seq_cap()
        seqs = np.zeros(len(bufs) * cap, dtype=SEQ_DT); metas = np.zeros(len(bufs), dtype=PARSE_DT)
        lstride = le.emu_lit_stride()
        lits = np.full(len(bufs) * lstride, 0xEE, dtype=np.uint8)
        emu_parse_units(le, src, units, seqs, lits, metas)
        for i, a in enumerate(bufs):
            cp = (C.c_uint * 7)(*cpv)
            n = len(a)
            oseq = np.zeros((n // 3 + 8, 3), dtype=np.uint32); olit = np.zeros(n + 64, dtype=np.uint8)
            litSize = C.c_size_t(0); rep = (C.c_uint * 3)()
            nb = lo.zo_parse_block(cp, _buf(a), n, _buf(oseq), len(oseq), _buf(olit), C.byref(litSize), rep)
            m = metas[i]
            s = seqs[i * cap: i * cap + int(m["nbSeq"])]
            assert int(m["nbSeq"]) == nb and int(m["litSize"]) == litSize.value, (cpv, cases[i][0])
            got = np.stack([s["litLength"].astype(np.uint32), s["mlBase"].astype(np.uint32) + 3, s["offBase"]], axis=1) if nb else np.zeros((0, 3), np.uint32)
            want = oseq[:nb].copy(); want[:, 0] &= 0xFFFF; want[:, 1] = ((want[:, 1] - 3) & 0xFFFF) + 3
            assert np.array_equal(got, want), (cpv, cases[i][0])
            assert np.array_equal(lits[i * lstride: i * lstride + litSize.value], olit[:litSize.value]), (cpv, cases[i][0])


def test_rowhash_parse_matches_oracle(libs):
    """strategies greedy / lazy / lazy2 with the ROW-HASH matcher (the reference's default for windowLog > 14): k_hc_chain builds the
    per-row links, k_hc_search_lds walks them with the tag filter, k_parse_lazy applies the 384-position skip rule and lazy
    skipping — against the oracle's row matcher (itself pinned to the reference with a fresh CCtx per unit)"""
    lo, le = libs
    lo.zo_set_row_matcher.argtypes = [C.c_int]
    lo.zo_set_row_matcher(1)
    try:
        for level in (5, 7, 10):                                # greedy, lazy, lazy2 (6 and 9 are the same strategies with other table sizes: GPU suite)
            cases = list(corpus_cases(lo, sizes=(131072, 40000), seeds=(1,)))
            rng = np.random.default_rng(level)
            big = np.tile(rng.integers(0, 256, 700, dtype=np.uint8), 190)[:131072].copy()       # matches of > 384 bytes: the skip rule
            big[rng.integers(0, 131072, 40)] ^= 0xFF
            cases.append(("period700", big))
            cases.append(("incompressible_then_text", np.concatenate([rng.integers(0, 256, 60000, dtype=np.uint8), cases[3][1][:71072]])))   # lazy skipping
            # every byte 24 times: each sequence a repcode (greedy takes it without a search, so nextToUpdate never moves and every batch start meets the whole gap
            # behind it: the 384-position rule flagged the unit's past again per batch until round 6 — 70 s here, 4.4 s per unit on the GPU)
            cases.append(("runs_of_24", np.repeat(rng.integers(0, 256, 131072 // 24 + 1, dtype=np.uint8), 24)[:131072]))
            bufs = [c[1] for c in cases]
            units = make_units(lo, [len(b) for b in bufs], level, row=True)
            assert (units["rowLog"] > 0).all()
            src = np.concatenate(bufs + [np.zeros(16, dtype=np.uint8)])
            cap = le.emu_seq_cap()
            seqs = np.zeros(len(bufs) * cap, dtype=SEQ_DT); metas = np.zeros(len(bufs), dtype=PARSE_DT)
            lstride = le.emu_lit_stride()
            lits = np.full(len(bufs) * lstride, 0xEE, dtype=np.uint8)
            emu_parse_units(le, src, units, seqs, lits, metas)
            for i, (name, a) in enumerate(cases):
                oseqs, litSize, rep, olits = oracle_parse(lo, a, level, want_lits=True)
                m = metas[i]
                s = seqs[i * cap: i * cap + int(m["nbSeq"])]
                ll = s["litLength"].astype(np.uint32); ml = s["mlBase"].astype(np.uint32) + 3
                if m["longType"] == 1: ll[m["longPos"]] += 0x10000
                if m["longType"] == 2: ml[m["longPos"]] += 0x10000
                got = np.stack([ll, ml, s["offBase"]], axis=1) if len(s) else np.zeros((0, 3), np.uint32)
                assert len(got) == len(oseqs), (level, name, len(got), len(oseqs))
                bad = np.nonzero((got != oseqs).any(axis=1))[0]
                assert len(bad) == 0, (level, name, int(bad[0]), got[bad[0]].tolist(), oseqs[bad[0]].tolist())
                assert int(m["litSize"]) == litSize and np.array_equal(lits[i * lstride: i * lstride + litSize], olits), (level, name)
                assert list(m["rep"][:2]) == rep[:2], (level, name)
    finally:
        lo.zo_set_row_matcher(0)


def test_order_sort_decides_the_global_table_wavefronts_by_mean_cost(libs):
    """k_order_sort also writes how many global-table workgroups join the queue (round 6): a batch whose mean estimated cost is below the
    threshold gets `gSparse`, a dense one keeps all that were launched (the word stays 0)"""
    lo, le = libs
    le.emu_order_sort_limit.restype = C.c_uint
    le.emu_order_sort_limit.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint]
    sparse = np.full(3000, 1900, dtype=np.uint32); dense = np.full(3000, 6500, dtype=np.uint32)
    mixed = np.concatenate([np.full(1500, 1000, dtype=np.uint32), np.full(1500, 9000, dtype=np.uint32)])       # mean 5 000
    assert le.emu_order_sort_limit(_buf(sparse), 3000, 1536, 4000) == 1536
    assert le.emu_order_sort_limit(_buf(dense), 3000, 1536, 4000) == 0
    assert le.emu_order_sort_limit(_buf(mixed), 3000, 1536, 4000) == 0
    assert le.emu_order_sort_limit(_buf(mixed), 3000, 1536, 5001) == 1536


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_queue_form_of_the_fast_stage_on_the_emulator(libs, monkeypatch, mode):
    """k_order_cost + k_order_sort + the persistent queue kernels (mode 1: LDS tables, 2: tables in global memory with ballot hash
    groups, 3: both kernels on one queue, 4: as 2 with all but one global-table workgroup told to stay out) give the oracle's
    sequences whatever order the units are taken in"""
    lo, le = libs
    monkeypatch.setenv("ZHIP_EMU_QUEUE", str(mode))
    for level in (1, -3):
        cases = []
        for n in (0, 7, 12, 13, 100, 1000, 5000, 40000, 131072):
            cases += list(corpus_cases(lo, sizes=(n,), seeds=(level + 11,)))
        cases = [c for c in cases if make_units(lo, [len(c[1])], level)["strategy"][0] == 1]
        check(le, lo, cases, level)


def test_rowhash_two_pass_prediction(libs, monkeypatch):
    """$ZHIP_RH_PREDICT=1: the parse is tried, units that had to redo more than a budget of searches live are parsed again after a predicting
    parse marked the positions the 384-position rule / lazy skipping will leave out and the records were recomputed without them
    (zhip_parse_lazy.h: rh_reconcile).  Same sequences as the oracle; far fewer live searches on long-match data."""
    lo, le = libs
    lo.zo_set_row_matcher.argtypes = [C.c_int]
    lo.zo_set_row_matcher(1)
    try:
        rng = np.random.default_rng(3)
        big = np.tile(rng.integers(0, 256, 700, dtype=np.uint8), 190)[:131072].copy()
        big[rng.integers(0, 131072, 40)] ^= 0xFF
        a = datagen(lo, 2 * 131072 + 5, 35, 2)
        bufs = [datagen(lo, 131072, 50, 1), a[:131072], a[131072:262144], a[262144:], big, text_like(60000, 2)]       # incl. a 5-byte unit
        for level in (5, 8):
            live = {}
            for mode in ("0", "1"):
                monkeypatch.setenv("ZHIP_RH_PREDICT", mode)
                units = make_units(lo, [len(b) for b in bufs], level, row=True)
                src = np.concatenate(bufs + [np.zeros(16, dtype=np.uint8)])
                cap = le.emu_seq_cap()
                seqs = np.zeros(len(bufs) * cap, dtype=SEQ_DT); metas = np.zeros(len(bufs), dtype=PARSE_DT)
                lstride = le.emu_lit_stride()
                lits = np.full(len(bufs) * lstride, 0xEE, dtype=np.uint8)
                emu_parse_units(le, src, units, seqs, lits, metas)
                for i, b in enumerate(bufs):
                    oseqs, litSize, rep, olits = oracle_parse(lo, b, level, want_lits=True)
                    m = metas[i]
                    assert int(m["status"]) == 0
                    s = seqs[i * cap: i * cap + int(m["nbSeq"])]
                    ll = s["litLength"].astype(np.uint32); ml = s["mlBase"].astype(np.uint32) + 3
                    if m["longType"] == 1: ll[m["longPos"]] += 0x10000
                    if m["longType"] == 2: ml[m["longPos"]] += 0x10000
                    got = np.stack([ll, ml, s["offBase"]], axis=1) if len(s) else np.zeros((0, 3), np.uint32)
                    assert len(got) == len(oseqs) and (got == oseqs).all(), (level, mode, i)
                    assert int(m["litSize"]) == litSize and np.array_equal(lits[i * lstride: i * lstride + litSize], olits), (level, mode, i)
                live[mode] = metas["pad0"].astype(np.int64)
            assert live["1"][0] * 3 < live["0"][0], (level, live)          # datagen P50: most searches were live, now few are
            assert live["1"].sum() < live["0"].sum()
    finally:
        lo.zo_set_row_matcher(0)


def test_rowhash_units_with_and_without_the_two_pass_prediction(libs, monkeypatch):
    """the unit kernels' live searches read the row lists (rh_live_lists; no live rows are kept for units since round 5) — with and without the units'
    two-pass prediction ($ZHIP_RH_PREDICT, the emulator driver's switch for what zhip_set_prediction does): the oracle's frames either way"""
    lo, le = libs
    bufs = [datagen(lo, 131072, 50, 4), datagen(lo, 100000, 80, 5), text_like(60000, 2)]
    lo.zo_set_row_matcher.argtypes = [C.c_int]
    lo.zo_set_row_matcher(1)
    try:
        want = []
        for b in bufs:
            dst = np.zeros(lo.zo_compress_bound(len(b)) + 64, dtype=np.uint8)
            r = lo.zo_compress_unit(_buf(dst), len(dst), _buf(b), len(b), 5)
            assert r != ERR
            want.append(dst[:r].tobytes())
    finally:
        lo.zo_set_row_matcher(0)
    for predict in ("0", "1"):
        monkeypatch.setenv("ZHIP_RH_PREDICT", predict)
        assert emu_compress_units(le, lo, bufs, 5, row=True) == want, predict


def test_lorem_ipsum_units_on_the_emulator(libs):
    """the product's two stages on the emulator, fed what `zstd -b#` benches without a file (LOREM_genBuffer(.., seed 0), programs/benchzstd.c:1014, made by the
    reference's own generator): frames equal the reference's, every match-finder family (row matcher for greedy / lazy / lazy2, fresh CCtx per unit)"""
    if not have_ref():
        pytest.skip("needs oracle/_ref (the reference build)")
    lo, le = libs
    lr = load_ref()
    lo.zo_set_row_matcher.argtypes = [C.c_int]
    a = lorem(lr, 131072 + 70000 + 4321, 0)
    bufs = [np.ascontiguousarray(a[:131072]), np.ascontiguousarray(a[131072:201072]), np.ascontiguousarray(a[201072:])]
    try:
        for level in (1, 3, 5, 7, 8):
            lo.zo_set_row_matcher(1 if level >= 5 else 0)
            frames = emu_compress_units(le, lo, bufs, level, row=level >= 5)
            for u, f in zip(bufs, frames):
                d = np.zeros(len(u) + 1024, dtype=np.uint8)
                k = lr.zref_compress_frame(level, _buf(u), len(u), _buf(d), len(d))
                assert k != ERR and bytes(f) == d[:k].tobytes(), (level, len(u))
    finally:
        lo.zo_set_row_matcher(0)
